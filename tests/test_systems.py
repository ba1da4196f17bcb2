"""Systems of PDEs: Lotka-Volterra (the workload of experiments/figure4.py) and SIR recipes, stacked kernels
(reference: pde/examples.py:84-248, pde/mixins.py:62-125,223-256, pde/problems.py:70-86, kernels.py:160-183).

CPU: the reference's own `test_pde_system` (tests/test_problems.py:165-208) restated for the host mirror; recipes and
Jacobians against the oracle and finite differences.  GPU: semilinear white-noise and latent-force EK1 on the systems
through the C ABI against the oracle."""

import numpy as np
import pytest
import scipy.linalg

import pnmol
import pnmol_oracle as o
from pnmol import kernels
from pnmol.pde import examples, problems


def test_pde_system_block_structure():
    """tests/test_problems.py:165-208."""
    pde1, pde2 = examples.heat_1d(bcond="neumann"), examples.heat_1d(bcond="neumann")
    pde = problems.SystemLinearPDENeumann(diffop=(pde1.diffop, pde2.diffop),
                                          diffop_scale=(pde1.diffop_scale, pde2.diffop_scale), bbox=pde1.bbox)
    assert pde.L is None and pde.E_sqrtm is None
    mesh = pnmol.mesh.RectangularMesh.from_bbox_1d([0.0, 1.0], step=0.1)
    kw = dict(mesh_spatial=mesh, kernel=kernels.SquareExponential(), stencil_size_interior=3, stencil_size_boundary=3)
    pde1.discretize(**kw), pde2.discretize(**kw)
    pde.discretize_system(**kw)
    for name in ("L", "E_sqrtm", "B", "R_sqrtm"):
        np.testing.assert_allclose(getattr(pde, name), scipy.linalg.block_diag(getattr(pde1, name), getattr(pde2, name)))


@pytest.mark.parametrize("name,ncomp", [("lotka_volterra_1d_discretized", 2), ("sir_1d_discretized", 3)])
def test_system_recipes_match_oracle(name, ncomp):
    """tests/test_problems.py:10-28 cases "sir", "lotka-volterra": shapes, and every matrix / function vs the oracle."""
    p, q = getattr(examples, name)(dx=0.2), getattr(o, name)(dx=0.2)
    N = p.mesh_spatial.shape[0]
    assert isinstance(p, problems.PDE) and np.isscalar(p.t0) and np.isscalar(p.tmax)
    assert p.L.shape == (ncomp * N, ncomp * N) and p.B.shape == (2 * ncomp, ncomp * N) and p.y0.shape == (ncomp * N,)
    for attr in ("L", "E_sqrtm", "B", "R_sqrtm", "y0"):
        np.testing.assert_allclose(getattr(p, attr), getattr(q, attr), rtol=1e-12, atol=1e-14)
    x = q.y0 * 1.01 + 0.3
    np.testing.assert_allclose(p.f(0.0, x), q.f(0.0, x))
    J = p.df(0.0, x)
    np.testing.assert_allclose(J, q.df(0.0, x))
    h = 1e-6
    Jn = np.stack([(p.f(0.0, x + h * e) - p.f(0.0, x - h * e)) / (2 * h) for e in np.eye(x.size)], axis=1)
    np.testing.assert_allclose(J, Jn, rtol=1e-6, atol=1e-7 * np.abs(J).max())      # closed form == what jax.jacfwd gives
    ivp = p.to_ivp()                                                              # mixins.py:195-214
    assert ivp.y0.shape == (ncomp * (N - 2),) and np.all(np.isfinite(ivp.f(p.t0, ivp.y0)))


def test_stacked_kernel_gram_is_block_diagonal():
    """kernels.py:160-183."""
    X = pnmol.mesh.RectangularMesh.from_bbox_1d([0.0, 1.0], step=0.25).points
    k = kernels.SquareExponential() + kernels.WhiteNoise()
    K1, K3 = k(X, X.T), kernels.duplicate(k, num=3)(X, X.T)
    np.testing.assert_allclose(K3, scipy.linalg.block_diag(K1, K1, K1))
    np.testing.assert_allclose(kernels.duplicate(k, num=3)(X, X), np.concatenate([k(X, X)] * 3))
    np.testing.assert_allclose(o.duplicate(o.SquareExponential() + o.WhiteNoise(), 3)(X, X.T), K3)


def _system_pair(name, ncomp, dx, dt, K, latent):
    kw = dict(dx=dx, tmax=K * dt)
    pde, opde = getattr(examples, name)(**kw), getattr(o, name)(**kw)
    k = kernels.duplicate(kernels.SquareExponential() + kernels.WhiteNoise(), num=ncomp)
    ok = o.duplicate(o.SquareExponential() + o.WhiteNoise(), ncomp)
    cls = pnmol.latent.SemiLinearLatentForceEK1 if latent else pnmol.white.SemiLinearWhiteNoiseEK1
    ocls = o.LatentForceEK1 if latent else o.WhiteNoiseEK1
    solver = cls(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
    osolver = ocls(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=True,
                   canonical_factor_signs=True)
    return pde, solver, opde, osolver


@pytest.mark.gpu
@pytest.mark.parametrize("name,ncomp,latent", [("lotka_volterra_1d_discretized", 2, False),
                                               ("lotka_volterra_1d_discretized", 2, True),
                                               ("sir_1d_discretized", 3, False)])
def test_system_solve_matches_oracle(name, ncomp, latent):
    """figure4.py:75-131 (Lotka-Volterra, white and latent) and the reference's deactivated SIR case
    (tests/test_pdefilter.py:96-137) on the GPU vs the oracle: north_star tolerances per component."""
    from helpers import assert_mean_std_parity
    dx, dt, K = 1.0 / 23, 2.0 ** -6, 6
    pde, solver, opde, osolver = _system_pair(name, ncomp, dx, dt, K, latent)
    sol, osol = solver.solve(pde), osolver.solve(opde)
    assert not np.isnan(sol.mean).any()
    d = pde.y0.shape[0]
    N = d // ncomp
    if latent:
        om, os_ = o.read_mean_and_std_latent(osol, osolver.state_iwp.projection_matrix(0))
    else:
        om, os_ = o.read_mean_and_std(osol, osolver.E0)
    m, s = sol.mean[:, 0, :d], sol.marginal_std[:, 0, :d]
    for c in range(ncomp):                       # components differ by orders of magnitude (SIR: 1e3 vs 1e0)
        sl = slice(c * N, (c + 1) * N)
        assert_mean_std_parity(m[:, sl], s[:, sl], om[:, sl], os_[:, sl])
