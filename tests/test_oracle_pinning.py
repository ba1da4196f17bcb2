"""Pin the CPU oracle with every known-answer / identity test the reference's own suite holds for the
hot path.  Each test names the reference test it restates."""

import numpy as np
import pytest
import scipy.integrate

import pnmol_oracle as o


# ---- tests/test_base/test_iwp.py ------------------------------------------------------------------
@pytest.fixture
def iwp():
    return o.IWP(wiener_process_dimension=1, num_derivatives=2, wp_diffusion_sqrtm=np.eye(1))


def test_iwp_closed_form_nu2(iwp):  # test_iwp.py:19-47
    dt = 0.1
    A, Ql = iwp.non_preconditioned_discretize(dt)
    np.testing.assert_allclose(A, [[1, dt, dt**2 / 2], [0, 1, dt], [0, 0, 1]], rtol=1e-13, atol=1e-15)
    Q = [[dt**5 / 20, dt**4 / 8, dt**3 / 6], [dt**4 / 8, dt**3 / 3, dt**2 / 2], [dt**3 / 6, dt**2 / 2, dt]]
    np.testing.assert_allclose(Ql @ Ql.T, Q, rtol=1e-12)


def test_iwp_preconditioner_identities(iwp):  # test_iwp.py:50-93
    dt = 0.1
    A, Ql = iwp.preconditioned_discretize
    An, Qn = iwp.non_preconditioned_discretize(dt)
    P, Pinv = iwp.nordsieck_preconditioner(dt)
    np.testing.assert_allclose(P @ A @ Pinv, An)
    np.testing.assert_allclose(P @ Ql, Qn)
    np.testing.assert_allclose(P @ Pinv, np.eye(3))
    s, sinv = iwp.nordsieck_preconditioner_1d_raw(dt)
    np.testing.assert_allclose(s * sinv, np.ones(3))
    np.testing.assert_allclose(A, [[1, 2, 1], [0, 1, 1], [0, 0, 1]])  # flip(pascal) -- SURVEY section 8 a8


def test_iwp_projection_matrix():  # test_iwp.py:99-114
    w = o.IWP(wiener_process_dimension=5, num_derivatives=2, wp_diffusion_sqrtm=np.eye(5))
    E0 = w.projection_matrix(0)
    assert E0.shape == (5, 15) and (E0 == 1).sum() == 5


# ---- tests/test_base/test_sqrt.py ------------------------------------------------------------------
@pytest.fixture(params=["full", "partial"])
def H_SQ_SC(request):
    H, SQ = o.IWP(1, 1, np.eye(1)).preconditioned_discretize_1d
    SC = SQ.copy()
    return (H, SQ, SC) if request.param == "full" else (H[:1], SQ[:1, :1], SC)


def test_propagate_cholesky_factor(H_SQ_SC):  # test_sqrt.py:36-45
    H, SQ, SC = H_SQ_SC
    chol = o.propagate_cholesky_factor(H @ SC, SQ)
    np.testing.assert_allclose(chol @ chol.T, H @ SC @ SC.T @ H.T + SQ @ SQ.T)
    np.testing.assert_allclose(np.tril(chol), chol)


@pytest.mark.parametrize("with_meascov", [True, False])
def test_update_sqrt_equals_classic(H_SQ_SC, with_meascov):  # test_sqrt.py:48-109
    H, SQ, SC = H_SQ_SC
    SC_new, K, Sl = o.update_sqrt(H, SC, SQ if with_meascov else None)
    S = H @ SC @ SC.T @ H.T + (SQ @ SQ.T if with_meascov else 0.0)
    Kref = SC @ SC.T @ H.T @ np.linalg.inv(S)
    np.testing.assert_allclose(SC_new @ SC_new.T, SC @ SC.T - Kref @ S @ Kref.T, atol=1e-14)
    np.testing.assert_allclose(SC_new, np.tril(SC_new))
    np.testing.assert_allclose(K, Kref)
    np.testing.assert_allclose(Sl @ Sl.T, S)
    np.testing.assert_allclose(Sl, np.tril(Sl))
    assert SC_new.shape == SC.shape and K.shape == (H.shape[1], H.shape[0])


# ---- tests/test_discretize.py ----------------------------------------------------------------------
def test_fd_weights_polynomial_kernel():  # test_discretize.py:52-71
    dx = 0.1
    mesh = o.RectMesh.from_bbox_1d([0.0, 1.0], step=dx)
    k = o.Polynomial(const=1.0)
    w, unc = o.fd_coefficients(mesh[1], mesh[[1, 0, 2]], k, k.laplace_x, k.laplace_xy)
    np.testing.assert_allclose(w * dx**2, [-2.0, 1.0, 1.0], atol=1e-8)
    np.testing.assert_allclose(unc, 0.0, atol=1e-8)


def test_fd_probabilistic_shapes_and_diagonal():  # test_discretize.py:74-101, :137-148
    mesh = o.RectMesh.from_bbox_1d([0.0, 1.0], step=0.1)
    L, E = o.fd_probabilistic_laplace(mesh)
    assert L.shape == E.shape == (11, 11)
    np.testing.assert_allclose(E, np.diag(np.diag(E)))
    B, R = o.fd_probabilistic_neumann_1d(mesh)
    assert B.shape == (2, 11) and R.shape == (2, 2)
    assert np.isfinite(L).all() and np.isfinite(B).all()


def test_kernel_derivatives_against_finite_differences():
    """The closed forms replacing jax autodiff (diffops.py:167-202)."""
    h = 1e-4
    for k in (o.SquareExponential(input_scale=1.3, output_scale=0.7), o.Matern52(input_scale=0.9),
              o.Polynomial(order=3, const=0.5)):
        x, y = np.array([0.31]), np.array([0.62])
        e = np.array([h])
        lap = (k.pair(x + e, y) - 2 * k.pair(x, y) + k.pair(x - e, y)) / h**2
        np.testing.assert_allclose(k.laplace_x(x, y), lap, rtol=1e-5)
        f = lambda yy: k.laplace_x(x, yy)  # noqa: E731
        lap2 = (f(y + e) - 2 * f(y) + f(y - e)) / h**2
        np.testing.assert_allclose(k.laplace_xy(x, y), lap2, rtol=1e-4)
        g = (k.pair(x + e, y) - k.pair(x - e, y)) / (2 * h)
        np.testing.assert_allclose(k.grad_x_1d(x, y), g, rtol=1e-6)
        gy = (k.grad_x_1d(x, y + e) - k.grad_x_1d(x, y - e)) / (2 * h)
        np.testing.assert_allclose(k.grad_xy_1d(x, y), gy, rtol=1e-6)
    # 2-d Laplacians of the SE / polynomial kernels
    for k in (o.SquareExponential(input_scale=0.8), o.Polynomial(order=4, const=1.0)):
        x, y = np.array([0.3, 0.1]), np.array([0.5, 0.45])
        lap = sum((k.pair(x + h * e, y) - 2 * k.pair(x, y) + k.pair(x - h * e, y)) / h**2 for e in np.eye(2))
        np.testing.assert_allclose(k.laplace_x(x, y), lap, rtol=1e-5)
    # Matern-5/2 at x == y: the Maclaurin values the reference hard-codes (discretize.py:184-197)
    m = o.Matern52(input_scale=1.7, output_scale=1.2)
    x = np.array([0.4])
    np.testing.assert_allclose(m.laplace_x(x, x), 1.7**2 * 1.2**2 * 2.5 / (1.0 - 2.5))
    np.testing.assert_allclose(m.laplace_xy(x, x), 1.2**2 * 1.7**4 * 3 * 2.5**2 / (2.0 - 3 * 2.5 + 2.5**2))


# ---- tests/test_problems.py ------------------------------------------------------------------------
@pytest.mark.parametrize("bcond", ["neumann", "dirichlet"])
def test_heat_ivp_jacobian_rows(bcond):  # test_problems.py:103-163
    dx, rate = 0.2, 0.01
    heat = o.heat_1d_discretized(dx=dx, bcond=bcond, kernel=o.Polynomial(), diffusion_rate=rate)
    n_in = heat.L.shape[0] - 2
    J = np.stack([heat.ivp_rhs(0.0, e) for e in np.eye(n_in)], axis=1)
    if bcond == "neumann":
        np.testing.assert_allclose(J[0, :2] * dx**2 / rate, [-1.0, 1.0], atol=1e-6)
    else:
        np.testing.assert_allclose(J, heat.L[1:-1, 1:-1])
    y0 = heat.y0[1:-1]
    np.testing.assert_allclose(J @ y0, heat.ivp_rhs(0.0, y0))
    assert heat.L.shape == heat.E_sqrtm.shape == (6, 6) and heat.B.shape[1] == 6


# ---- tests/test_mesh.py ----------------------------------------------------------------------------
def test_mesh_facts():  # test_mesh.py:22-97
    pts = np.array([[0.0, 0.0, -1.0], [0.1, 0.2, 0.3], [0.6, 0.21, 0.672], [1.0, 1.0, 2.0], [0.0, 1.0, 0.0]])
    mesh = o.RectMesh(pts)
    np.testing.assert_allclose(mesh.bbox, [[0.0, 1.0], [0.0, 1.0], [-1.0, 2.0]])
    assert list(mesh.boundary[1]) == [True, False, False, True, True]
    assert list(mesh.interior[1]) == [False, True, True, False, False]
    nb, _ = mesh.neighbours(pts[0], num=2)
    np.testing.assert_allclose(nb, pts[[0, 1]])
    g1 = o.RectMesh.from_bbox_1d([0.0, 1.0], step=0.1)
    assert g1.shape == (11, 1)
    B = g1.boundary_projection_matrix
    np.testing.assert_array_equal(B @ g1.points, g1.boundary[0])
    g2 = o.RectMesh.from_bbox_2d(np.array([[0.0, 0.0], [1.0, 1.0]]), steps=(0.1, 0.1))
    assert g2.shape == (121, 2)
    np.testing.assert_array_equal(g2.boundary_projection_matrix @ g2.points, g2.boundary[0])
    for N in (32, 256, 512, 1024):  # int(1/dx)+1 gives exactly N
        assert o.RectMesh.from_bbox_1d([0.0, 1.0], step=1.0 / (N - 1)).shape[0] == N


# ---- tests/test_odetools/test_step.py ---------------------------------------------------------------
def test_step_rules():  # test_step.py:15-122
    c = o.Constant(0.1)
    assert c.suggest(np.nan, 0.1) == 0.1 and c.is_accepted(0.1) and c.scale_error_estimate(None, None) is None
    assert c.first_dt(None) == 0.1
    a = o.Adaptive(abstol=0.1, reltol=0.01)
    assert a.is_accepted(0.99) and not a.is_accepted(1.01)
    assert a.suggest(0.3, 0.5, local_convergence_rate=2) > 0.3 > a.suggest(0.3, 2.0, local_convergence_rate=2)
    err, ref = np.array([0.5, 0.6]), np.array([2.0, 3.0])
    np.testing.assert_allclose(a.scale_error_estimate(err, ref), np.linalg.norm(err / (0.1 + 0.01 * ref)) / np.sqrt(2))
    np.testing.assert_allclose(a.scale_error_estimate(err[:1], ref[:1]), err[0] / (0.1 + 0.01 * ref[0]))
    assert a.first_dt(o.heat_1d_discretized()) > 0.0
    with pytest.raises(ValueError):
        a.suggest(0.3, 0.5)


# ---- tests/test_pdefilter.py -------------------------------------------------------------------------
@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_solve_smoke_problem(bcond):  # test_pdefilter.py:15-64, :141-146
    pde = o.heat_1d_discretized(tmax=1.0, dx=0.2, diffusion_rate=0.05, kernel=o.SquareExponential(), bcond=bcond)
    solver = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(0.1),
                             spatial_kernel=o.SquareExponential() + o.WhiteNoise())
    sol = solver.solve(pde)
    assert not np.isnan(sol.mean).any() and not np.isnan(sol.cov_sqrtm).any()
    # sum of ten 0.1 lands one ulp short of 1.0 -> a runt 11th step
    assert sol.info["num_steps"] == sol.info["num_attempted_steps"] == 11 and len(sol.t) == 12
    assert sol.mean.shape == (12, 3, 6) and sol.cov_sqrtm.shape == (12, 18, 18)
    final, info = solver.simulate_final_state(pde)
    np.testing.assert_allclose(final.y.cov_sqrtm, sol.cov_sqrtm[-1] * np.sqrt(sol.diffusion_squared_calibrated))


# ---- independent cross-checks -----------------------------------------------------
@pytest.mark.parametrize("N,nu,bcond", [(24, 2, "dirichlet"), (24, 1, "neumann"), (48, 2, "dirichlet")])
def test_sqrt_form_equals_covariance_form(N, nu, bcond):
    """The reference's QR filter and the classic covariance filter agree on mean and covariance when both
    start from the same state (licence for the GPU's covariance form, test_sqrt.py:48-78 at full scale)."""
    dt = 2.0**-7
    pde = o.heat_1d_discretized(tmax=20 * dt, dx=1.0 / (N - 1), kernel=o.SquareExponential(), bcond=bcond)
    s = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(dt), spatial_kernel=o.Matern52() + o.WhiteNoise())
    state = s.initialize(pde)
    mean, cov = state.y.mean, state.y.cov_sqrtm @ state.y.cov_sqrtm.T
    for _ in range(20):
        new, _ = s.attempt_step(state, dt, pde)
        mean, cov, sig2, err = o.covariance_form_step(s, pde, mean, cov, dt, state.t)
        C = new.y.cov_sqrtm
        np.testing.assert_allclose(mean, new.y.mean, rtol=1e-8, atol=1e-10 * np.abs(new.y.mean).max())
        np.testing.assert_allclose(np.diag(cov), np.einsum("ij,ij->i", C, C), rtol=1e-6, atol=1e-9 * np.abs(cov).max())
        np.testing.assert_allclose(err, new.error_estimate, rtol=1e-8)
        state = new


def test_filter_mean_tracks_method_of_lines_solution():
    """Procedure of experiments/figure1.py:57-73: the posterior mean stays within a few calibrated standard
    deviations of a fine solve_ivp solution of y' = L y (zero Dirichlet padding)."""
    N, dt, K = 24, 2.0**-6, 64
    pde = o.heat_1d_discretized(tmax=K * dt, dx=1.0 / (N - 1), kernel=o.SquareExponential(), diffusion_rate=0.05)
    s = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=o.Matern52() + o.WhiteNoise())
    sol = s.solve(pde)
    means, stds = o.calibrated_mean_and_std(sol, s.E0)
    ref = scipy.integrate.solve_ivp(pde.ivp_rhs, (0.0, K * dt), pde.y0[1:-1], t_eval=sol.t, rtol=1e-10, atol=1e-12)
    truth = np.pad(ref.y.T, ((0, 0), (1, 1)))
    err = np.abs(means - truth)
    assert err.max() < 1e-4 and err.max() < 2e-3 * np.abs(truth).max()
    # after the initial transient the error is covered by the calibrated uncertainty
    assert np.all(err[K // 2:, 1:-1] < 5 * stds[K // 2:, 1:-1] + 1e-9)


def test_sigma2_quirk_is_sign_dependent():
    """Quirk Q1: the reference's `diffusion_squared_local` (white.py:125) changes when the rows of the
    QR factor R1 are sign-flipped -- a mathematically irrelevant choice LAPACK makes from the data."""
    dt = 2.0**-7
    pde = o.heat_1d_discretized(tmax=4 * dt, dx=1.0 / 31, kernel=o.SquareExponential())
    kw = dict(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=o.Matern52() + o.WhiteNoise())
    a, b = o.WhiteNoiseEK1(**kw), o.WhiteNoiseEK1(canonical_factor_signs=True, **kw)
    sa, sb = a.solve(pde), b.solve(pde)
    np.testing.assert_allclose(sa.mean, sb.mean, rtol=1e-12, atol=1e-18)
    assert abs(sa.diffusion_squared_calibrated / sb.diffusion_squared_calibrated - 1.0) > 1e-3
