"""Latent-force EK1 (scope row f3; reference: src/pnmol/latent.py, base/stacked_ssm.py).

CPU: the oracle's restatement against the only fixture the reference holds for this solver (its no-NaN smoke
test, tests/test_pdefilter.py:140-145 -- numerically the oracle stays UNPINNED for this row) and against an
independent covariance-form restatement; the product's closed-form initialisation against the oracle's.
GPU: the product through the C ABI against the oracle."""

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from pnmol import kernels


def _pair(N, nu, dt, K, bcond, semilinear=False, dx=None):
    dx = 1.0 / (N - 1) if dx is None else dx
    kw = dict(tmax=K * dt, dx=dx, diffusion_rate=0.05, bcond=bcond, stencil_size_interior=3,
              stencil_size_boundary=3, nugget_gram_matrix_fd=0.0)
    if semilinear:
        pde = pnmol.pde.examples.spruce_budworm_1d_discretized(kernel=kernels.SquareExponential(), **kw)
        opde = o.spruce_budworm_1d_discretized(kernel=o.SquareExponential(), **kw)
        cls = pnmol.latent.SemiLinearLatentForceEK1
    else:
        pde = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.SquareExponential(), **kw)
        opde = o.heat_1d_discretized(kernel=o.SquareExponential(), **kw)
        cls = pnmol.latent.LinearLatentForceEK1
    solver = cls(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
                 spatial_kernel=kernels.SquareExponential() + kernels.WhiteNoise())
    osolver = o.LatentForceEK1(num_derivatives=nu, steprule=o.Constant(dt), semilinear=semilinear,
                               spatial_kernel=o.SquareExponential() + o.WhiteNoise(), canonical_factor_signs=True)
    return pde, solver, opde, osolver


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
@pytest.mark.parametrize("semilinear", [False, True])
def test_oracle_latent_solve_no_nan(bcond, semilinear):
    """tests/test_pdefilter.py:46-53,:87-94,:140-145 (case_linear_latent / case_semilinear_latent)."""
    _, _, opde, osolver = _pair(6, 2, 0.1, 10, bcond, semilinear, dx=0.2)
    sol = osolver.solve(opde)
    assert sol.mean.shape[1:] == (3, 12)
    assert not np.isnan(sol.mean).any() and not np.isnan(sol.cov_sqrtm).any()


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_oracle_latent_matches_covariance_form(bcond):
    _, _, opde, osolver = _pair(12, 2, 0.05, 6, bcond)
    sol = osolver.solve(opde)
    st = osolver.initialize(opde)
    mean, cov, t = st.y.mean, st.y.cov_sqrtm @ st.y.cov_sqrtm.T, opde.t0
    for k in range(1, 6):
        mean, cov, _ = o.latent_covariance_form_step(osolver, opde, mean, cov, 0.05, t)
        t += 0.05
        ref = sol.cov_sqrtm[k] @ sol.cov_sqrtm[k].T
        np.testing.assert_allclose(mean, sol.mean[k], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(cov, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())


def test_stacked_ssm_matches_oracle_blocks():
    """base/stacked_ssm.py:17-80 against the oracle's block_diag assembly."""
    _, solver, opde, osolver = _pair(6, 2, 0.1, 2, "dirichlet")
    osolver.initialize(opde)
    pde = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.SquareExponential(), tmax=0.2, dx=0.2,
                                                 diffusion_rate=0.05, bcond="dirichlet")
    st, lf, E0, E1, _ = solver.initialize_iwp_latent(pde)
    ssm = pnmol.base.stacked_ssm.StackedSSM([st, lf])
    assert ssm.state_dimension == 2 * 3 * 6
    import scipy.linalg
    A, Q = ssm.preconditioned_discretize
    (As, Qs), (Ae, Qe) = osolver.state_iwp.preconditioned_discretize, osolver.lf_iwp.preconditioned_discretize
    np.testing.assert_allclose(A, scipy.linalg.block_diag(As, Ae))
    np.testing.assert_allclose(Q, scipy.linalg.block_diag(Qs, Qe), atol=1e-14)
    P, Pi = ssm.nordsieck_preconditioner(0.1)
    np.testing.assert_allclose(P @ Pi, np.eye(36), atol=1e-12)
    A2, Q2 = ssm.non_preconditioned_discretize(0.1)
    np.testing.assert_allclose(A2, P @ A @ Pi, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ssm.projection_matrix(0, 1) @ np.arange(36.0), 18 + 3 * np.arange(6.0))
    np.testing.assert_allclose(ssm.projection_matrix(1), scipy.linalg.block_diag(E1, E1))


@pytest.mark.parametrize("N,nu,bcond,semilinear", [(6, 2, "dirichlet", False), (24, 2, "neumann", False),
                                                   (24, 1, "dirichlet", False), (24, 2, "dirichlet", True)])
def test_latent_structured_initialisation_matches_oracle(N, nu, bcond, semilinear):
    """latent.py:20-134 block-wise in closed form on the host (cold path) = the reference's two QR updates."""
    pde, s, opde, so = _pair(N, nu, 0.1, 2, bcond, semilinear)
    s.state_iwp, s.lf_iwp, s.E0, s.E1, gamma = s.initialize_iwp_latent(pde)
    s._gram, s._gram_latent = gamma @ gamma.T, pde.E_sqrtm @ pde.E_sqrtm.T
    mean, blocks = s._initial_moments(pde)
    st = so.initialize(opde)
    cov = st.y.cov_sqrtm @ st.y.cov_sqrtm.T
    n, D = nu + 1, (nu + 1) * N
    np.testing.assert_allclose(mean[:2], st.y.mean[:2], rtol=1e-6, atol=1e-8 * np.abs(st.y.mean).max())
    # derivatives >= 2 (and >= 1 of eps) are independent of both data sets: exactly 0 here, while the QR form
    # leaves rounding noise amplified by the 1e-6 nugget (measured 1.2e-6 at N=24)
    assert np.all(mean[2:] == 0.0) and np.abs(st.y.mean[2:]).max(initial=0.0) < 1e-5 * np.abs(st.y.mean).max()
    seen = np.zeros_like(cov, dtype=bool)
    for (pa, a, pb, b), blk in blocks.items():
        sl = (slice(pa * D + a, (pa + 1) * D, n), slice(pb * D + b, (pb + 1) * D, n))
        ref = cov[sl]
        seen[sl] = True
        tol = 1e-6 if (pa, a) == (pb, b) else 1e-3
        np.testing.assert_allclose(blk, ref, rtol=tol, atol=tol * max(np.abs(ref).max(), 1e-12))
    assert np.abs(cov[~seen]).max() <= 1e-9 * np.abs(cov).max()   # every other block of the reference's C0 is zero


# ------------------------------------------------------------------------------------------------ GPU
def _oracle_marginals(osolver, sol, d):
    E0 = osolver.state_iwp.projection_matrix(0)
    return o.read_mean_and_std_latent(sol, E0)


@pytest.mark.gpu
@pytest.mark.parametrize("N,nu,bcond,K", [(6, 2, "dirichlet", 10), (24, 2, "neumann", 8), (40, 1, "dirichlet", 8),
                                          (40, 3, "dirichlet", 6), (96, 2, "dirichlet", 6)])
def test_latent_solve_matches_oracle(N, nu, bcond, K):
    """`LinearLatentForceEK1.solve` through the C ABI vs the oracle: glued means, marginal stds (state AND latent
    force), local diffusions.  north_star tolerances (mean rtol 1e-5, std rtol 1e-4, floors as in helpers)."""
    from helpers import assert_mean_std_parity
    dt = 2.0 ** -6
    pde, solver, opde, osolver = _pair(N, nu, dt, K, bcond)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    assert sol.mean.shape == osol.mean.shape == (K + 1, nu + 1, 2 * N)
    np.testing.assert_allclose(sol.t, osol.t, rtol=0, atol=1e-14)
    ovar = np.einsum("tij,tij->ti", osol.cov_sqrtm, osol.cov_sqrtm)
    n = nu + 1
    ostd = np.sqrt(np.stack([np.hstack((v[:n * N].reshape((n, N), order="F"), v[n * N:].reshape((n, N), order="F")))
                             for v in ovar]))
    std = sol.marginal_std
    for a in range(n):                                           # per derivative (scales differ by orders of magnitude)
        for half in (slice(0, N), slice(N, 2 * N)):
            assert_mean_std_parity(sol.mean[:, a, half], std[:, a, half], osol.mean[:, a, half], ostd[:, a, half])
    m, s = o.read_mean_and_std_latent(osol, osolver.state_iwp.projection_matrix(0))
    np.testing.assert_allclose(sol.mean[:, 0, :N], m, rtol=1e-5, atol=1e-5 * np.abs(m).max())


@pytest.mark.gpu
def test_latent_solve_marginals_on_device_loop():
    """The constant-step loop kept on the device (`pnmol_filter_steps`) returns the state half's read-out
    (experiments/figure1.py:83-89) and the same local diffusions as the oracle (canonical factor signs)."""
    from helpers import assert_mean_std_parity
    N, nu, dt, K = 64, 2, 2.0 ** -6, 12
    pde, solver, opde, osolver = _pair(N, nu, dt, K, "dirichlet")
    t, means, stds, sig, final = solver.solve_marginals(pde)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std_latent(osol, osolver.state_iwp.projection_matrix(0))
    assert means.shape == om.shape == (K + 1, N)
    assert_mean_std_parity(means, stds, om, os_)
    ost = None
    d2 = []
    for st, _ in osolver.solution_generator(opde):
        if not isinstance(st.diffusion_squared_local, list):
            d2.append(st.diffusion_squared_local)
    np.testing.assert_allclose(sig, np.array(d2), rtol=1e-4)
    assert final.y.mean.shape == (nu + 1, 2 * N)


@pytest.mark.gpu
@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_semilinear_latent_matches_oracle(bcond):
    """`SemiLinearLatentForceEK1` (spruce budworm, tests/test_pdefilter.py:87-94) vs the oracle."""
    from helpers import assert_mean_std_parity
    N, nu, dt, K = 32, 2, 2.0 ** -6, 8
    pde, solver, opde, osolver = _pair(N, nu, dt, K, bcond, semilinear=True)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    om, os_ = o.read_mean_and_std_latent(osol, osolver.state_iwp.projection_matrix(0))
    assert not np.isnan(sol.mean).any()
    assert_mean_std_parity(sol.mean[:, 0, :N], sol.marginal_std[:, 0, :N], om, os_)


@pytest.mark.gpu
@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_reference_smoke_case_latent(bcond):
    """The reference's own test for this solver, run on the product: tests/test_pdefilter.py:46-53,:140-145."""
    pde = pnmol.pde.examples.heat_1d_discretized(tmax=1.0, dx=0.2, stencil_size_interior=3, stencil_size_boundary=3,
                                                 diffusion_rate=0.05, kernel=kernels.SquareExponential(),
                                                 nugget_gram_matrix_fd=0.0, bcond=bcond)
    solver = pnmol.latent.LinearLatentForceEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt=0.1),
                                               spatial_kernel=kernels.SquareExponential() + kernels.WhiteNoise())
    solution = solver.solve(pde)
    assert not np.any(np.isnan(solution.mean))
    assert not np.any(np.isnan(solution.cov_sqrtm))
