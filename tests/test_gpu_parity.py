"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle.  Run with -m gpu."""

import numpy as np
import pytest
import scipy.linalg

import pnmol_oracle as oracle
from helpers import assert_mean_std_parity, make_pair, to_device_layout

DT = 2.0 ** -7

pytestmark = pytest.mark.gpu


def _oracle_internals(osolver, opde, state, dt):
    """Intermediate quantities of one oracle step in the Nordsieck frame (white.py:97-123)."""
    P, Pinv = osolver.iwp.nordsieck_preconditioner(dt)
    A, Ql = osolver.iwp.preconditioned_discretize
    m = Pinv @ state.y.mean.reshape(-1, order="F")
    C = Pinv @ state.y.cov_sqrtm
    mp = A @ m
    Pm = A @ C @ C.T @ A.T + Ql @ Ql.T
    z, H, E = osolver.evaluate_ode(opde, osolver.E0 @ P, osolver.E1 @ P, mp, state.t + dt)
    S = H @ Pm @ H.T + E @ E.T
    Ls = np.linalg.cholesky(S)
    W = scipy.linalg.solve_triangular(Ls, H @ Pm, lower=True).T
    r = scipy.linalg.solve_triangular(Ls, z, lower=True)
    return dict(mp=mp, Pm=Pm, z=z, S=S, Ls=Ls, W=W, r=r)


@pytest.mark.parametrize("N,nu,bcond", [(32, 2, "dirichlet"), (20, 1, "neumann"), (40, 3, "dirichlet")])
def test_step_stages(hip_ctx, N, nu, bcond):
    """Every stage of one step against the oracle: P-, z, Ls, W, r, then the posterior."""
    dt = 2.0 ** -7
    pde, solver, opde, osolver = make_pair(N, nu, dt, 4, bcond)
    ostate = osolver.initialize(opde)
    solver.initialize(pde)                       # binds the device model
    flt = solver._device_filter
    n, d = nu + 1, N
    dims = flt.dims()
    dp, mp, m = dims["dp"], dims["mp"], dims["m"]
    dev = flt.new_state()
    dev.set(0.0, ostate.y.mean, ostate.y.cov_sqrtm @ ostate.y.cov_sqrtm.T)
    solver._ensure_error_model(pde, dt)
    out, info, err = flt.step(dev, dt)
    ref = _oracle_internals(osolver, opde, ostate, dt)

    Dp = n * dp
    Ppred = flt.debug_read(0, Dp * Dp).reshape(Dp, Dp)
    np.testing.assert_allclose(Ppred, to_device_layout(ref["Pm"], n, d, dp), rtol=1e-12, atol=1e-13 * np.abs(ref["Pm"]).max())
    z = flt.debug_read(4, mp)
    np.testing.assert_allclose(z[:m], ref["z"], rtol=1e-10, atol=1e-12 * np.abs(ref["z"]).max())
    assert np.all(z[m:] == 0)
    rows = mp + Dp + 32
    F = flt.debug_read(2, rows * mp).reshape(rows, mp)
    np.testing.assert_allclose(F[:m, :m], ref["Ls"], rtol=1e-8, atol=1e-10 * np.abs(ref["Ls"]).max())
    Wdev = F[mp:mp + Dp, :m]
    Wref = np.zeros((Dp, m))
    for a in range(n):
        Wref[a * dp:a * dp + d] = ref["W"][a::n]
    np.testing.assert_allclose(Wdev, Wref, rtol=1e-7, atol=1e-9 * np.abs(Wref).max())
    np.testing.assert_allclose(F[mp + Dp, :m], ref["r"], rtol=1e-7, atol=1e-9 * np.abs(ref["r"]).max())

    onew, _ = osolver.attempt_step(ostate, dt, opde)
    np.testing.assert_allclose(out.mean(), onew.y.mean, rtol=1e-8, atol=1e-10 * np.abs(onew.y.mean).max())
    ocov = onew.y.cov_sqrtm @ onew.y.cov_sqrtm.T
    np.testing.assert_allclose(out.cov(), ocov, rtol=1e-6, atol=1e-9 * np.abs(ocov).max())
    np.testing.assert_allclose(out.marginal_var().reshape(-1, order="F"), np.diag(ocov), rtol=1e-6,
                               atol=1e-9 * np.abs(ocov).max())
    assert info.info == -1
    np.testing.assert_allclose(info.diffusion_squared_local, onew.diffusion_squared_local, rtol=1e-7)
    np.testing.assert_allclose(info.sigma2_whitened, ref["r"] @ ref["r"] / m, rtol=1e-8)
    np.testing.assert_allclose(err, onew.error_estimate, rtol=1e-7)


@pytest.mark.parametrize("N,nu,K,bcond", [(32, 1, 100, "dirichlet"), (64, 2, 100, "dirichlet"),
                                          (64, 2, 40, "neumann"), (256, 2, 8, "dirichlet")])
def test_solve_marginals_parity(hip_ctx, N, nu, K, bcond):
    """BASELINE configs 0/1 (+ Neumann): K steps of dt=2^-7, mean rtol 1e-5, std rtol 1e-4."""
    dt = 2.0 ** -7
    pde, solver, opde, osolver = make_pair(N, nu, dt, K, bcond)
    t, means, stds, sig, final = solver.solve_marginals(pde)
    osol = osolver.solve(opde)
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert len(t) == K + 1 and np.array_equal(t, osol.t)
    assert_mean_std_parity(means, stds, omeans, ostds)
    # calibrated read-out of figure1.py:19-23 with the sign-canonical diffusion
    np.testing.assert_allclose(np.mean(sig), osol.diffusion_squared_calibrated, rtol=1e-6)


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_solve_api_smoke_problem(hip_ctx, bcond):
    """The reference's own smoke case (tests/test_pdefilter.py:15-64,141-146): N=6, dt=0.1, tmax=1 ->
    11 steps incl. the runt step; here also checked numerically against the oracle."""
    pde, solver, opde, osolver = make_pair(6, 2, 0.1, 10, bcond, dx=0.2)
    pde.tmax = opde.tmax = 1.0
    solver.spatial_kernel = __import__("pnmol").kernels.SquareExponential() + __import__("pnmol").kernels.WhiteNoise()
    osolver.spatial_kernel = oracle.SquareExponential() + oracle.WhiteNoise()
    sol = solver.solve(pde)
    osol = osolver.solve(opde)
    assert sol.info == osol.info and sol.info["num_steps"] == 11
    assert np.array_equal(sol.t, osol.t)
    assert not np.any(np.isnan(sol.mean)) and not np.any(np.isnan(sol.cov_sqrtm))
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    # steps 0..10: the north-star tolerances.  Step 11 is the runt step (dt = 1.1e-16): with Nordsieck scales
    # of 1e+-40 the reference algorithm itself only resolves it to ~1e-3 (the oracle's result loses the mirror
    # symmetry of the problem at that level; the reference's own test only asserts "no NaN"), so it is
    # compared loosely.
    assert_mean_std_parity(sol.mean[:-1, 0], sol.marginal_std[:-1, 0], omeans[:-1], ostds[:-1])
    np.testing.assert_allclose(sol.mean[:-1], osol.mean[:-1], rtol=1e-5, atol=1e-5 * np.abs(osol.mean).max())
    np.testing.assert_allclose(sol.mean[-1, 0], omeans[-1], rtol=0, atol=2e-3 * np.abs(omeans).max())
    np.testing.assert_allclose(sol.marginal_std[-1, 0], ostds[-1], rtol=0, atol=5e-2 * np.abs(ostds).max())
    # the factor is not unique; C C^T is
    cov = sol.cov_sqrtm @ np.transpose(sol.cov_sqrtm, (0, 2, 1))
    ocov = osol.cov_sqrtm @ np.transpose(osol.cov_sqrtm, (0, 2, 1))
    np.testing.assert_allclose(cov[:-1], ocov[:-1], rtol=1e-4, atol=1e-7 * np.abs(ocov).max())
    np.testing.assert_allclose(np.mean([sol.diffusion_squared_calibrated]), osol.diffusion_squared_calibrated, rtol=0.2)


def test_attempt_step_is_functional(hip_ctx):
    """attempt_step must not modify its input state (the driver re-uses it after a rejection,
    pdefilter.py:192-223)."""
    pde, solver, _, _ = make_pair(24, 2, 2.0 ** -6, 4)
    s0 = solver.initialize(pde)
    before = s0.y.cov.copy()
    a, _ = solver.attempt_step(s0, 2.0 ** -6, pde)
    b, _ = solver.attempt_step(s0, 2.0 ** -6, pde)
    assert np.array_equal(s0.y.cov, before)
    assert np.array_equal(a.y.mean, b.y.mean) and np.array_equal(a.y.cov, b.y.cov)
    c, _ = solver.attempt_step(s0, 2.0 ** -7, pde)      # a different dt from the same state
    assert not np.array_equal(a.y.mean, c.y.mean)


@pytest.mark.parametrize("N,K", [(512, 3), (1024, 2)])
def test_large_mesh_steps(hip_ctx, N, K):
    """BASELINE headline size (N=512) and config 2 (N=1024), nu=2: a few steps against the oracle
    (the oracle needs ~1.5 s resp. ~12 s per step at these sizes), then size-independent properties
    over a longer run: symmetry of the covariance, monotone time grid, finite calibrated diffusion."""
    dt = 2.0 ** -7
    pde, solver, opde, osolver = make_pair(N, 2, dt, K)
    t, means, stds, sig, final = solver.solve_marginals(pde)
    osol = osolver.solve(opde)
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert np.array_equal(t, osol.t)
    assert_mean_std_parity(means, stds, omeans, ostds)
    assert np.all(np.isfinite(sig)) and np.all(sig > 0)
    if N == 512:
        cov = final.y.cov
        np.testing.assert_allclose(cov, cov.T, rtol=0, atol=1e-12 * np.abs(cov).max())
        var = final.y.marginal_var.reshape(-1, order="F")
        np.testing.assert_allclose(np.diag(cov), var, rtol=1e-12, atol=0)
        ocov = osol.cov_sqrtm[-1] @ osol.cov_sqrtm[-1].T
        np.testing.assert_allclose(cov, ocov, rtol=1e-3, atol=1e-6 * np.abs(ocov).max())


def test_two_dimensional_mesh(hip_ctx):
    """BASELINE config 4's shape at a size the oracle can do: 2-d Dirichlet heat problem (12x12 mesh, 5-point
    stencils, nu=1).  The reference ships no 2-d recipe; both sides assemble it from the reference's parts."""
    import pnmol

    dt, K = 2.0 ** -8, 12
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
    opde = oracle.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=oracle.SquareExponential())
    for name in ("L", "B", "y0"):
        np.testing.assert_allclose(getattr(pde, name), getattr(opde, name), rtol=1e-12, atol=1e-14)
    solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                             spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    t, means, stds, sig, _ = solver.solve_marginals(pde)
    osol = osolver.solve(opde)
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert means.shape == (K + 1, 144)
    assert_mean_std_parity(means, stds, omeans, ostds)


def test_state_roundtrip_and_clone(hip_ctx):
    """C-ABI state functions: set/get in the reference's F-order, clone, argument checking."""
    import ctypes
    from pnmol import _hip

    pde, solver, opde, osolver = make_pair(20, 2, 2.0 ** -6, 2, "neumann")
    solver.initialize(pde)
    flt = solver._device_filter
    rng = np.random.default_rng(0)
    D = 60
    A = rng.standard_normal((D, D))
    cov, mean = A @ A.T, rng.standard_normal((3, 20))
    st = flt.new_state()
    st.set(0.25, mean, cov)
    assert st.t == 0.25
    np.testing.assert_array_equal(st.mean(), mean)
    np.testing.assert_array_equal(st.cov(), cov)
    np.testing.assert_array_equal(st.marginal_var().reshape(-1, order="F"), np.diag(cov))
    cl = st.clone()
    np.testing.assert_array_equal(cl.cov(), cov)
    lib = flt.lib
    assert lib.pnmol_filter_step(flt.handle, st.handle, 0.1, st.handle, None, None) == -1      # aliasing
    assert lib.pnmol_filter_step(flt.handle, st.handle, -1.0, cl.handle, None, None) == -1     # dt <= 0
    assert lib.pnmol_filter_steps(flt.handle, st.handle, 0, 0.1, None, None, None) == -1        # k < 1
    with pytest.raises(_hip.PnmolHipError):
        bad = flt.new_state()
        bad.set(0.0, mean, -cov)                      # a non-PSD covariance must be reported, not swallowed
        flt.step(bad, 2.0 ** -6)


def test_full_size_2d_mesh_properties(hip_ctx):
    """BASELINE config 5's mesh (64x64 Dirichlet heat problem, nu=1: D=8192, m=4348) in fp64.  The oracle needs
    minutes per step at this size, so only size-independent properties are checked: finite, symmetric PSD-diagonal
    covariance, boundary nodes pinned, mean decaying like the analytic heat mode, determinism of a repeated run."""
    import pnmol

    dt, K, kappa = 2.0 ** -9, 3, 0.05
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(64, 64), tmax=K * dt, diffusion_rate=kappa,
                                                           kernel=pnmol.kernels.SquareExponential())
    solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                             spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    t, means, stds, sig, final = solver.solve_marginals(pde)
    assert means.shape == (K + 1, 4096) and np.all(np.isfinite(means)) and np.all(np.isfinite(stds))
    assert np.all(np.isfinite(sig)) and np.all(sig > 0)
    on_boundary = pde.mesh_spatial.boundary[1]
    assert np.abs(means[-1][on_boundary]).max() < 1e-8
    assert stds[-1][on_boundary].max() < 1e-4 * stds[-1].max()
    # y0 = 0.1 sin(pi x) sin(pi y) is an eigenmode: u(t) = exp(-2 kappa pi^2 t) y0 up to the FD error
    expected = np.exp(-2 * kappa * np.pi ** 2 * t[-1]) * pde.y0
    np.testing.assert_allclose(means[-1], expected, rtol=0, atol=2e-3 * np.abs(pde.y0).max())
    var = final.y.marginal_var
    assert var.min() > -1e-9 * var.max()
    t2, means2, stds2, sig2, _ = solver.solve_marginals(pde)
    assert np.array_equal(means, means2) and np.array_equal(stds, stds2) and np.array_equal(sig, sig2)


def test_concurrent_problems_on_one_gpu():
    """Several problems in flight on ONE GPU (one context / stream each, `steps_begin` .. `steps_end`): the sweep kernel's
    workgroups wait for each other through global-memory flags, so kernels of different problems sharing the CUs must
    neither deadlock nor disturb each other -- every problem must reproduce its own sequential result bit for bit."""
    import pnmol
    from pnmol import _hip, batch
    dt, K, B = 2.0 ** -7, 12, 6
    runs = []
    for g in range(B):
        pde = pnmol.pde.examples.heat_1d_discretized(tmax=K * dt, dx=1.0 / 255, diffusion_rate=batch.diffusion_sweep(g, 8),
                                                     kernel=pnmol.kernels.SquareExponential(), bcond="dirichlet")
        solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt),
                                                 spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
        solver._context = _hip.Context(0)
        state = solver.initialize(pde)
        solver._ensure_error_model(pde, dt)
        flt, dev = solver._device_filter, state.y.device_state
        ref = flt.new_state()
        ref.set(pde.t0, dev.mean(), dev.cov())
        seq = flt.steps(ref, K, dt)                      # sequential reference, this problem alone on the device
        runs.append((solver, flt, dev, seq))
    for _, flt, dev, _ in runs:
        flt.steps_begin(dev, K, dt)
    for _, flt, dev, (m_seq, s_seq, i_seq) in runs:
        m, s, infos = flt.steps_end(dev)
        assert all(o.info == -1 for o in infos)
        assert np.array_equal(m, m_seq) and np.array_equal(s, s_seq)
        assert [o.diffusion_squared_local for o in infos] == [o.diffusion_squared_local for o in i_seq]


@pytest.mark.parametrize("N,bcond", [(40, "dirichlet"), (96, "neumann")])
def test_device_error_model_matches_host(hip_ctx, N, bcond):
    """`pnmol_filter_prepare_error_model` (Sq factorised on the GPU by the step's own kernels) against the host
    O(m^3) solve of `_error_model` and against the oracle's `estimate_error` (white.py:153-162)."""
    dt = 2.0 ** -6
    pde, solver, opde, osolver = make_pair(N, 2, dt, 3, bcond=bcond)
    s0 = solver.initialize(pde)
    a, _ = solver.attempt_step(s0, dt, pde)                      # device error model (default)
    solver.error_model_on_host = True
    solver._device_filter.error_model_dt = None
    b, _ = solver.attempt_step(s0, dt, pde)
    np.testing.assert_allclose(a.error_estimate, b.error_estimate, rtol=1e-8)
    o0 = osolver.initialize(opde)
    oa, _ = osolver.attempt_step(o0, dt, opde)
    np.testing.assert_allclose(a.error_estimate, oa.error_estimate, rtol=1e-5, atol=1e-9 * np.abs(oa.error_estimate).max())


@pytest.mark.parametrize("N,splits", [(40, (1, 2, 5)), (96, (3, 4, 1, 1, 6))])
def test_loop_steps_match_single_steps(hip_ctx, N, splits):
    """Inside `pnmol_filter_steps` the step boundary is moved (2 launches per step: the down-date epilogue predicts the
    next covariance in place, the posterior covariance is only written by the call's last step).  Any split of K steps
    into calls must reproduce K self-contained `pnmol_filter_step`s: read-outs, scalars, final covariance -- up to the
    rounding of the predict, which the two paths contract into FMAs differently (observed: 1e-20 on entries of size
    1e-2)."""
    dt = 2.0 ** -6
    K = sum(splits)
    pde, solver, _, _ = make_pair(N, 2, dt, K)
    s0 = solver.initialize(pde)
    flt = solver._device_filter
    solver._ensure_error_model(pde, dt)
    single = flt.new_state()
    single.set(pde.t0, s0.y.mean, s0.y.cov)
    ref_means, ref_sig = [], []
    cur = single
    for _ in range(K):
        cur, info, _ = flt.step(cur, dt)
        ref_means.append(cur.mean()[0].copy())
        ref_sig.append(info.diffusion_squared_local)
    loop = flt.new_state()
    loop.set(pde.t0, s0.y.mean, s0.y.cov)
    means, sig = [], []
    for k in splits:
        m, s, infos = flt.steps(loop, k, dt)
        means.extend(m)
        sig.extend(o.diffusion_squared_local for o in infos)
        assert np.array_equal(loop.mean()[0], m[-1])          # the state is complete after every call
    ref_means = np.array(ref_means)
    np.testing.assert_allclose(np.array(means), ref_means, rtol=1e-10, atol=1e-14 * np.abs(ref_means).max())
    np.testing.assert_allclose(sig, ref_sig, rtol=1e-9)
    c_ref = cur.cov()
    np.testing.assert_allclose(loop.cov(), c_ref, rtol=1e-7, atol=1e-11 * np.abs(c_ref).max())
    np.testing.assert_allclose(loop.marginal_var(), cur.marginal_var(), rtol=1e-7, atol=1e-11 * np.abs(c_ref).max())


@pytest.mark.parametrize("N,nu,bcond", [(20, 2, "neumann"), (40, 1, "dirichlet"), (64, 2, "dirichlet")])
def test_cov_sqrtm_is_the_cholesky_factor(hip_ctx, N, nu, bcond):
    """`state.y.cov_sqrtm` (pnmol_state_get_cov_sqrtm: device Cholesky of the covariance in the reference's state
    order) is lower triangular with non-negative diagonal, reproduces the covariance, and equals the oracle's QR factor
    (base/sqrt.py:33-73) after fixing the latter's column signs -- where the factor is well determined (columns whose
    pivot is not at the rounding level of the covariance)."""
    dt, K = 2.0 ** -6, 4
    pde, solver, opde, osolver = make_pair(N, nu, dt, K, bcond=bcond)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    C = sol.cov_sqrtm[-1]
    cov = sol._ys[-1].cov
    assert np.array_equal(C, np.tril(C)) and np.all(np.diag(C) >= 0.0)
    # A pivot below 1e-13 of its diagonal entry (or negative: the covariance is PSD only up to the rounding of the
    # recursion) is dropped (zero column).  Its mass stays in the trailing matrix, so diagonals are accurate (1e-6); only the off-diagonal entries of a dropped direction j are lost, and those are
    # bounded by sqrt(c_ii * 1e-13 c_jj) (Schur complements of a PSD matrix).
    E = C @ C.T - cov
    dv = np.sqrt(np.maximum(np.diag(cov), 0.0))
    assert np.all(np.abs(E) <= 1e-6 * np.outer(dv, dv) + 1e-12 * np.abs(cov).max())
    np.testing.assert_allclose(np.diag(C @ C.T), np.diag(cov), rtol=1e-6, atol=1e-13 * np.abs(cov).max())
    Co = osol.cov_sqrtm[-1]
    Co = Co * np.where(np.diag(Co) < 0, -1.0, 1.0)[None, :]              # column signs of the QR factor
    ocov = Co @ Co.T
    np.testing.assert_allclose(cov, ocov, rtol=1e-3, atol=1e-7 * np.abs(ocov).max())
    # Columns whose pivot is significant on the scale of its derivative class (index c % n in the F order).  Where the
    # pivot is lost (noise-free boundary nodes and what they determine) the factorisation is not unique: the QR factor
    # keeps an arbitrary direction there, the Cholesky factor a zero column -- same C C^T.
    n = nu + 1
    dvar = np.diag(ocov)
    cls_max = np.array([dvar[a::n].max() for a in range(n)])
    well = np.diag(Co) ** 2 > 1e-4 * cls_max[np.arange(dvar.size) % n]
    first_bad = np.flatnonzero(~well)[0] if (~well).any() else well.size
    if bcond == "neumann":                      # (Dirichlet: the very first node is noise-free -> nothing is unique)
        assert first_bad >= n                   # columns behind a lost pivot inherit its arbitrariness
    for c in range(first_bad):
        np.testing.assert_allclose(C[:, c], Co[:, c], rtol=0, atol=1e-3 * np.abs(Co[:, c]).max())


def test_stop_at_adjusts_the_step_grid(hip_ctx):
    """`solve(pde, stop_at=...)` (pdefilter.py:140-160, `_TimeStopper` :238-256): steps are cut at the requested times and
    the solver hits them exactly; same time grid and numbers as the oracle's driver."""
    dt = 2.0 ** -6
    pde, solver, opde, osolver = make_pair(24, 2, dt, 6)
    stops = [1.5 * dt, 3.25 * dt]
    sol = solver.solve(pde, stop_at=stops)
    osol = osolver.solve(opde, stop_at=stops)
    assert np.array_equal(sol.t, osol.t)
    assert all(any(abs(t - s) < 1e-15 for t in sol.t) for s in stops)
    assert sol.info == osol.info
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(sol.mean[:, 0], sol.marginal_std[:, 0], omeans, ostds)


@pytest.mark.parametrize("N", [96, 256])
def test_filters_alive_together_agree_bit_for_bit(hip_ctx, N):
    """The first live filter of a process runs the sweep in its XCD-local layout (chain workgroup and S row blocks on one
    XCD, hand-over through its L2), later ones in the spread layout (write-through hand-over): the arithmetic is the same,
    so three filters of one problem that are alive together must agree bit for bit -- whichever layouts they got."""
    import gc
    gc.collect()
    dt, K = 2.0 ** -7, 12
    runs, keep = [], []
    for _ in range(3):
        pde, solver, _, _ = make_pair(N, 2, dt, K)
        keep.append(solver)                      # (keeps the device filter alive)
        runs.append(solver.solve_marginals(pde))
    for r in runs[1:]:
        assert np.array_equal(runs[0][1], r[1]) and np.array_equal(runs[0][2], r[2]) and np.array_equal(runs[0][3], r[3])


@pytest.mark.parametrize("N", [256, 1024])
def test_stale_hand_over_buffers_are_never_read(hip_ctx, N):
    """the cross-workgroup hand-over of the sweep kernels relies on "a tile is only read behind its flag, and
    the first touch after the kernel-start invalidate is fresh".  A reader that took a line from an earlier launch (round 2
    had one: 8-byte sc1 loads served from the XCD's L2) would go unnoticed while that launch wrote nearly the same numbers.
    Here every hand-over buffer (F, the L_jj^-1 tiles, the feed / scratch tiles) is filled with NaN by another launch between
    calls: the steps must give the same bits as without it.  N=256: k_sweep_rl; N=1024: k_sweep."""
    import pnmol
    K = 6
    pde, solver, _, _ = make_pair(N, 2, DT, 2 * K)
    state = solver.initialize(pde)
    solver._ensure_error_model(pde, DT)
    flt, dev = solver._device_filter, state.y.device_state
    clean, dirty = flt.new_state(), flt.new_state()
    for st in (clean, dirty):
        st.set(pde.t0, dev.mean(), dev.cov())
    m1, s1, i1 = flt.steps(clean, K, DT)                    # the same call sequence without the poison
    o1, info1, err1 = flt.step(clean, DT)
    m1b, s1b, i1b = flt.steps(o1, K - 1, DT)
    flt.debug_poison()
    m2, s2, i2 = flt.steps(dirty, K, DT)
    flt.debug_poison()
    o, info, err = flt.step(dirty, DT)                      # the self-contained single step as well
    flt.debug_poison()
    m2b, s2b, i2b = flt.steps(o, K - 1, DT)
    assert np.all(np.isfinite(m2)) and np.all(np.isfinite(s2)) and all(x.info == -1 for x in i2)
    assert np.array_equal(m1, m2) and np.array_equal(s1, s2)
    assert [x.diffusion_squared_local for x in i1] == [x.diffusion_squared_local for x in i2]
    assert np.array_equal(o1.mean(), o.mean()) and np.array_equal(err1, err)
    assert info1.diffusion_squared_local == info.diffusion_squared_local
    assert np.array_equal(m1b, m2b) and np.array_equal(s1b, s2b)
    assert [x.diffusion_squared_local for x in i1b] == [x.diffusion_squared_local for x in i2b]


def test_sixteen_problems_in_flight():
    """a dependency of the sweep kernel may point to a workgroup that has no CU yet (the chain workgroup
    waits for the feed of a higher block index; dispatch order holds per XCD only), so with many sweeps in flight forward
    progress rests on the bounded spins and on workgroups retiring.  Sixteen problems (twice the benchmark's batch) in
    flight on one device must finish without a timed-out wait (info == -1 everywhere) and bit-identical to their serial runs."""
    import pnmol
    from pnmol import _hip, batch
    K, B = 10, 16
    runs = []
    for g in range(B):
        pde = pnmol.pde.examples.heat_1d_discretized(tmax=K * DT, dx=1.0 / 255, diffusion_rate=batch.diffusion_sweep(g % 8, 8),
                                                     kernel=pnmol.kernels.SquareExponential(), bcond="dirichlet")
        solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(DT),
                                                 spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
        solver._context = _hip.Context(0)
        state = solver.initialize(pde)
        solver._ensure_error_model(pde, DT)
        flt, dev = solver._device_filter, state.y.device_state
        ref = flt.new_state()
        ref.set(pde.t0, dev.mean(), dev.cov())
        runs.append((solver, flt, dev, flt.steps(ref, K, DT)))
    for _, flt, dev, _ in runs:
        flt.steps_begin(dev, K, DT)
    for _, flt, dev, (m_seq, s_seq, i_seq) in runs:
        m, s, infos = flt.steps_end(dev)
        assert all(o.info == -1 for o in infos)
        assert np.array_equal(m, m_seq) and np.array_equal(s, s_seq)
