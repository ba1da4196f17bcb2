"""The filter step in square-root (QR) form on the GPU (`pnmol.sqrtform`, include/pnmol_sqrt.h) against the oracle's
restatement of the reference algorithm as written (white.py:96-146 with base/sqrt.py): same state (mean, cov_sqrtm),
same sequence of two QRs, so the FACTOR itself is compared (up to the column signs LAPACK leaves arbitrary)."""

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from helpers import assert_mean_std_parity, make_pair

pytestmark = pytest.mark.gpu


def _canon(C):
    return C * np.where(np.diag(C) < 0, -1.0, 1.0)[None, :]


def _sqrt_solver(nu, dt, semilinear=False, kernel=None):
    cls = pnmol.sqrtform.SemiLinearWhiteNoiseEK1 if semilinear else pnmol.sqrtform.LinearWhiteNoiseEK1
    return cls(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
               spatial_kernel=kernel or pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())


@pytest.mark.parametrize("N,nu,bcond", [(32, 2, "dirichlet"), (32, 1, "neumann"), (50, 3, "neumann"), (128, 2, "dirichlet"),
                                        (512, 2, "neumann")])   # the headline size: three tree levels in the update QR
def test_single_step_from_the_oracles_state(N, nu, bcond):
    """One attempt_step from the oracle's own initial state: mean, factor, sigma^2."""
    dt = 2.0 ** -7
    pde, _, opde, osolver = make_pair(N, nu, dt, 4, bcond=bcond)
    solver = _sqrt_solver(nu, dt)
    solver.initialize(pde)
    ostate = osolver.initialize(opde)
    state = pnmol.pdefilter.PDEFilterState(t=ostate.t, y=pnmol.base.rv.MultivariateNormal(ostate.y.mean, ostate.y.cov_sqrtm),
                                           error_estimate=None, reference_state=None, diffusion_squared_local=[])
    for _ in range(3):
        new, _ = solver.attempt_step(state, dt, pde)
        onew, _ = osolver.attempt_step(ostate, dt, opde)
        np.testing.assert_allclose(new.y.mean, onew.y.mean, rtol=1e-7, atol=1e-9 * np.abs(onew.y.mean).max())
        C, oC = new.y.cov_sqrtm, _canon(onew.y.cov_sqrtm)
        assert np.all(np.triu(C, 1) == 0) and np.all(np.diag(C) >= 0)
        cov, ocov = C @ C.T, oC @ oC.T
        sd = np.sqrt(np.diag(ocov))
        sd = np.maximum(sd, 1e-8 * sd.max())      # exactly-zero variances (Dirichlet nodes) are 1e-21-level noise in both
        assert np.max(np.abs(cov - ocov) / np.outer(sd, sd)) < 1e-6
        # the factor itself, where it is determined: a noise-free Dirichlet node comes first in the state order with a
        # pivot of 1e-21, and every later column of ANY triangular factor then carries cov[:, 0] / 1e-21 = rounding
        # noise of O(1e-4): only C C^T is defined there (checked above); with Neumann conditions the factor is unique
        if bcond == "neumann":
            np.testing.assert_allclose(C, oC, rtol=1e-5, atol=1e-8 * np.abs(oC).max())
        np.testing.assert_allclose(new.diffusion_squared_local, onew.diffusion_squared_local, rtol=1e-6)
        state, ostate = new, onew


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_solve_matches_oracle(bcond):
    N, nu, dt, K = 40, 2, 2.0 ** -6, 10
    pde, _, opde, osolver = make_pair(N, nu, dt, K, bcond=bcond)
    solver = _sqrt_solver(nu, dt)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(sol.mean[:, 0], sol.marginal_std[:, 0], om, os_)
    np.testing.assert_allclose(sol.diffusion_squared_calibrated, osol.diffusion_squared_calibrated, rtol=1e-5)
    assert sol.cov_sqrtm.shape == osol.cov_sqrtm.shape
    # square-root form resolves the small standard deviations the covariance form floors at 1e-5 max(std)
    big = os_[1:] > 1e-9 * os_.max()
    np.testing.assert_allclose(sol.marginal_std[1:, 0][big], os_[1:][big], rtol=1e-6)


def test_device_loop_matches_stepwise_solve_and_covariance_form():
    N, nu, dt, K = 64, 2, 2.0 ** -7, 8
    pde, cov_solver, _, _ = make_pair(N, nu, dt, K)
    solver = _sqrt_solver(nu, dt)
    t, means, stds, sig, final = solver.solve_marginals(pde)
    sol = _sqrt_solver(nu, dt).solve(pde)
    np.testing.assert_allclose(means, sol.mean[:, 0], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(stds[1:], sol.marginal_std[1:, 0], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(np.mean(sig), sol.diffusion_squared_calibrated, rtol=1e-9)
    tc, mc, sc, sigc, _ = cov_solver.solve_marginals(pde)
    assert np.array_equal(t, tc)
    assert_mean_std_parity(mc, sc, means, stds)
    np.testing.assert_allclose(sigc, sig, rtol=1e-5)


def test_semilinear_spruce_budworm():
    dt, K = 2.0 ** -4, 6
    pde = pnmol.pde.examples.spruce_budworm_1d_discretized(dx=0.1, tmax=K * dt)
    opde = o.spruce_budworm_1d_discretized(dx=0.1, tmax=K * dt)
    k, ok = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise(), o.SquareExponential() + o.WhiteNoise()
    sol = _sqrt_solver(2, dt, semilinear=True, kernel=k).solve(pde)
    osolver = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=True,
                              canonical_factor_signs=True)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(sol.mean[:, 0], sol.marginal_std[:, 0], om, os_)


def test_structured_and_dense_predict_agree():
    """A non-triangular factor of the same covariance takes the dense member lists in the predict QR, a triangular one the
    structured lists: same step result."""
    N, nu, dt = 45, 2, 2.0 ** -7
    pde, _, _, _ = make_pair(N, nu, dt, 2, bcond="neumann")
    solver = _sqrt_solver(nu, dt)
    s0 = solver.initialize(pde)
    C = np.asarray(s0.y.cov_sqrtm)
    assert np.all(np.triu(C, 1) == 0)
    a, _ = solver.attempt_step(s0, dt, pde)
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal(C.shape))
    s1 = s0._replace(y=pnmol.base.rv.MultivariateNormal(s0.y.mean, C @ Q))     # same covariance, dense factor
    b, _ = solver.attempt_step(s1, dt, pde)
    np.testing.assert_allclose(a.y.mean, b.y.mean, rtol=1e-9, atol=1e-12 * np.abs(a.y.mean).max())
    np.testing.assert_allclose(a.y.cov_sqrtm, b.y.cov_sqrtm, rtol=1e-6, atol=1e-9 * np.abs(a.y.cov_sqrtm).max())


# ---- latent-force model in square-root form (latent.py:20-233) -------------------------------------------------------
@pytest.mark.parametrize("bcond,semilinear", [("neumann", False), ("dirichlet", False), ("neumann", True)])
def test_latent_solve_matches_oracle(bcond, semilinear):
    N, nu, dt, K = 24, 2, 2.0 ** -6, 6
    k, ok = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise(), o.SquareExponential() + o.WhiteNoise()
    if semilinear:
        pde = pnmol.pde.examples.spruce_budworm_1d_discretized(dx=1.0 / (N - 1), tmax=K * dt)
        opde = o.spruce_budworm_1d_discretized(dx=1.0 / (N - 1), tmax=K * dt)
        cls = pnmol.sqrtform.SemiLinearLatentForceEK1
    else:
        kw = dict(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=0.05, bcond=bcond)
        pde = pnmol.pde.examples.heat_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **kw)
        opde = o.heat_1d_discretized(kernel=o.SquareExponential(), **kw)
        cls = pnmol.sqrtform.LinearLatentForceEK1
    solver = cls(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
    osolver = o.LatentForceEK1(num_derivatives=nu, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=semilinear,
                               canonical_factor_signs=True)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    d = pde.L.shape[0]
    # glued means and marginal stds of BOTH halves (u and eps), every derivative
    # (the noise-free update with 1e-6 nuggets is conditioned ~1e10: two Householder QRs agree to ~1e-7 of the scale)
    np.testing.assert_allclose(sol.mean, osol.mean, rtol=1e-5, atol=1e-6 * np.abs(osol.mean).max())
    ovar = np.einsum("tij,tij->ti", osol.cov_sqrtm, osol.cov_sqrtm)
    var = np.einsum("tij,tij->ti", sol.cov_sqrtm, sol.cov_sqrtm)
    np.testing.assert_allclose(np.sqrt(var), np.sqrt(ovar), rtol=1e-4, atol=1e-6 * np.sqrt(ovar).max())
    om, os_ = o.read_mean_and_std_latent(osol, osolver.state_iwp.projection_matrix(0))
    assert_mean_std_parity(sol.mean[:, 0, :d], sol.marginal_std[:, 0, :d], om[:, :d], os_[:, :d])
    np.testing.assert_allclose(sol.diffusion_squared_calibrated, osol.diffusion_squared_calibrated, rtol=1e-5)


def test_latent_device_loop():
    N, nu, dt, K = 24, 2, 2.0 ** -6, 5
    kw = dict(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=0.05, bcond="neumann", kernel=pnmol.kernels.SquareExponential())
    pde = pnmol.pde.examples.heat_1d_discretized(**kw)
    k = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise()
    mk = lambda: pnmol.sqrtform.LinearLatentForceEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
    t, means, stds, sig, final = mk().solve_marginals(pde)
    sol = mk().solve(pde)
    np.testing.assert_allclose(means, sol.mean[:, 0, :N], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(stds[1:], sol.marginal_std[1:, 0, :N], rtol=1e-8, atol=1e-14)


# ---- estimate_error in square-root form: the Adaptive rule (odetools/step.py:58-119) ---------------------------------
@pytest.mark.parametrize("semilinear", [False, True])
def test_adaptive_steps_follow_the_oracle(semilinear):
    kw = dict(abstol=1e-3, reltol=1e-2)
    k, ok = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise(), o.SquareExponential() + o.WhiteNoise()
    if semilinear:
        pkw = dict(tmax=0.4, dx=1.0 / 11, diffusion_rate=0.05, bcond="neumann")
        pde = pnmol.pde.examples.spruce_budworm_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **pkw)
        opde = o.spruce_budworm_1d_discretized(kernel=o.SquareExponential(), **pkw)
        cls = pnmol.sqrtform.SemiLinearWhiteNoiseEK1
    else:
        pkw = dict(tmax=0.4, dx=1.0 / 11, diffusion_rate=0.05, bcond="neumann")
        pde = pnmol.pde.examples.heat_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **pkw)
        opde = o.heat_1d_discretized(kernel=o.SquareExponential(), **pkw)
        cls = pnmol.sqrtform.LinearWhiteNoiseEK1
    solver = cls(num_derivatives=2, spatial_kernel=k, steprule=pnmol.odetools.step.Adaptive(**kw))
    osolver = o.WhiteNoiseEK1(num_derivatives=2, spatial_kernel=ok, semilinear=semilinear, canonical_factor_signs=True,
                              steprule=o.Adaptive(**kw))
    sol, osol = solver.solve(pde), osolver.solve(opde)
    assert sol.info == osol.info and sol.info["num_attempted_steps"] >= sol.info["num_steps"] > 3
    np.testing.assert_allclose(sol.t, osol.t, rtol=1e-9)
    np.testing.assert_allclose(sol.mean[:, 0], osol.mean[:, 0], rtol=1e-6, atol=1e-8 * np.abs(osol.mean).max())


def test_error_estimate_matches_oracle():
    N, nu, dt = 40, 2, 2.0 ** -6
    pde, _, opde, osolver = make_pair(N, nu, dt, 2, bcond="dirichlet")
    solver = pnmol.sqrtform.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Adaptive(),
                                                spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s, os_ = solver.initialize(pde), osolver.initialize(opde)
    new, _ = solver.attempt_step(s, dt, pde)
    onew, _ = osolver.attempt_step(os_, dt, opde)
    np.testing.assert_allclose(new.error_estimate, onew.error_estimate, rtol=1e-6)


# ---- at BASELINE's full size: two independent device algorithms for the same step ------------------------------------
def test_full_size_covariance_form_agrees_with_square_root_form():
    """N=512, nu=2, dt=2^-7 (BASELINE config 1), 25 steps: the covariance form (Cholesky sweep + down-date, the bench
    path) against the square-root form (two Householder QRs per step).  They share the problem assembly and nothing
    else; agreement at the north-star tolerances is a full-size check no CPU oracle run is needed for."""
    N, nu, dt, K = 512, 2, 2.0 ** -7, 25
    pde, cov_solver, _, _ = make_pair(N, nu, dt, K)
    t, mq, sq, sigq, _ = _sqrt_solver(nu, dt).solve_marginals(pde)
    tc, mc, sc, sigc, _ = cov_solver.solve_marginals(pde)
    assert np.array_equal(t, tc) and len(t) == K + 1
    assert np.isfinite(mq).all() and np.isfinite(sq).all()
    assert_mean_std_parity(mc, sc, mq, sq)
    np.testing.assert_allclose(mc, mq, rtol=1e-8, atol=1e-10 * np.abs(mq).max())      # means agree far below the bar
    np.testing.assert_allclose(sigc, sigq, rtol=1e-6)


# ---- the reference's initialisation on the device, for the covariance-form solvers (white.initialize_on_device) -------
@pytest.mark.parametrize("N,nu,bcond", [(32, 2, "dirichlet"), (40, 1, "neumann"), (128, 2, "dirichlet")])
def test_device_initialisation_matches_oracle_and_host(N, nu, bcond):
    dt = 2.0 ** -7
    pde, solver, opde, osolver = make_pair(N, nu, dt, 4, bcond=bcond)
    host = solver.initialize(pde)
    solver.initialize_on_device = True
    dev = solver.initialize(pde)
    ost = osolver.initialize(opde)
    ocov = ost.y.cov_sqrtm @ ost.y.cov_sqrtm.T
    scale = np.sqrt(np.abs(np.diag(ocov)).max())
    for st in (dev, host):
        np.testing.assert_allclose(st.y.mean, ost.y.mean, rtol=1e-7, atol=1e-9 * np.abs(ost.y.mean).max())
        cov = st.y.cov
        sd = np.maximum(np.sqrt(np.abs(np.diag(ocov))), 1e-8 * scale)
        assert np.max(np.abs(cov - ocov) / np.outer(sd, sd)) < 1e-5
    # and the solve that starts from it
    sol = solver.solve(pde)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(sol.mean[:, 0], sol.marginal_std[:, 0], om, os_)


def test_one_qr_step_equals_the_two_qr_step(monkeypatch):
    """The step-invariant rows factored once + one structured QR per step (default from the second step on) against the
    reference's two QRs per step (PNMOL_SQRT_ONE_QR=0): the same R, so the same solve."""
    N, nu, dt, K = 70, 2, 2.0 ** -7, 6
    pde, _, _, _ = make_pair(N, nu, dt, K, bcond="neumann")
    monkeypatch.setenv("PNMOL_SQRT_ONE_QR", "0")
    t2, m2, s2, sig2, f2 = _sqrt_solver(nu, dt).solve_marginals(pde)
    monkeypatch.delenv("PNMOL_SQRT_ONE_QR")
    t1, m1, s1, sig1, f1 = _sqrt_solver(nu, dt).solve_marginals(pde)
    np.testing.assert_allclose(m1, m2, rtol=1e-9, atol=1e-12 * np.abs(m2).max())
    np.testing.assert_allclose(s1, s2, rtol=1e-7, atol=1e-12 * s2.max())
    np.testing.assert_allclose(sig1, sig2, rtol=1e-8)
    C1, C2 = f1.y.cov_sqrtm, f2.y.cov_sqrtm
    np.testing.assert_allclose(C1, C2, rtol=1e-6, atol=1e-9 * np.abs(C2).max())


@pytest.mark.parametrize("env", [{"PNMOL_QR_FUSE": "0"}, {"PNMOL_QR_INLOOP": "0"}, {"PNMOL_QR_OWNER": "0"}, {"PNMOL_QR_PRE": "1"}])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_qr_launch_sequences_agree(monkeypatch, env, dtype):
    """The launch sequences of the device QR (include/pnmol_sqrt.h; switches documented at qr_launch_panel): the default --
    trailing update of a tree level and panel factorisation of the next level in one launch, V^T V and T formed inside the column
    loop -- against each A/B variant, three tree levels (N = 150: 113 row blocks in the first update QR) and two.  Same
    reflectors up to the order of sums: the solves agree to rounding (fp32 QR: to fp32 rounding)."""
    N, nu, dt, K = 150, 2, 2.0 ** -7, 4
    pde, _, _, _ = make_pair(N, nu, dt, K, bcond="neumann")

    def run():
        s = _sqrt_solver(nu, dt)
        s.dtype = dtype
        return s.solve_marginals(pde)

    t1, m1, s1, sig1, f1 = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t2, m2, s2, sig2, f2 = run()
    tol = 1e-9 if dtype == "f64" else 2e-5
    np.testing.assert_allclose(m1, m2, rtol=tol, atol=tol * np.abs(m2).max())
    np.testing.assert_allclose(s1, s2, rtol=100 * tol, atol=tol * s2.max())


def test_gather_products_equal_the_dense_products(monkeypatch):
    """H has short rows (stencil + derivative entry): (H T1)^T and R H^T are gathers over an ELL image of H by default;
    PNMOL_SQRT_ELL=0 keeps the dense MFMA products.  Same sums up to their order."""
    N, nu, dt, K = 70, 2, 2.0 ** -7, 5
    pde, _, _, _ = make_pair(N, nu, dt, K, bcond="neumann")
    monkeypatch.setenv("PNMOL_SQRT_ELL", "0")
    t2, m2, s2, sig2, f2 = _sqrt_solver(nu, dt).solve_marginals(pde)
    monkeypatch.delenv("PNMOL_SQRT_ELL")
    t1, m1, s1, sig1, f1 = _sqrt_solver(nu, dt).solve_marginals(pde)
    np.testing.assert_allclose(m1, m2, rtol=1e-9, atol=1e-12 * np.abs(m2).max())
    np.testing.assert_allclose(s1, s2, rtol=1e-7, atol=1e-12 * s2.max())
    np.testing.assert_allclose(sig1, sig2, rtol=1e-8)


def test_two_dimensional_mesh_square_root_form():
    """2-d Dirichlet heat problem (5-point stencils, nu=1; the shape of BASELINE config 5): 12x12 against the oracle,
    28x28 (D=1568, m=892: three tree levels, a boundary block of 108 rows) against the covariance form on the GPU."""
    dt, K = 2.0 ** -8, 6
    k = pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise()
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
    opde = o.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=o.SquareExponential())
    osolver = o.WhiteNoiseEK1(num_derivatives=1, steprule=o.Constant(dt), canonical_factor_signs=True,
                              spatial_kernel=o.Matern52() + o.WhiteNoise())
    t, means, stds, sig, _ = _sqrt_solver(1, dt, kernel=k).solve_marginals(pde)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(means, stds, om, os_)
    np.testing.assert_allclose(np.mean(sig), osol.diffusion_squared_calibrated, rtol=1e-5)

    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(28, 28), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
    tq, mq, sq, sigq, _ = _sqrt_solver(1, dt, kernel=k).solve_marginals(pde)
    cov = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
    tc, mc, sc, sigc, _ = cov.solve_marginals(pde)
    assert_mean_std_parity(mc, sc, mq, sq)
    np.testing.assert_allclose(sigc, sigq, rtol=1e-5)
