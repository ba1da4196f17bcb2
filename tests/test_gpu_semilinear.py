"""SemiLinearWhiteNoiseEK1 (white.py:189-208) on the GPU vs the oracle; also the Adaptive step rule through
perform_full_step (pdefilter.py:177-227).  Run with -m gpu."""

import numpy as np
import pytest

import pnmol
import pnmol_oracle as oracle
from helpers import assert_mean_std_parity

pytestmark = pytest.mark.gpu


def _pair(N, dx, dt, tmax, bcond, nu=2, steprule=None, osteprule=None, prior="se"):
    kw = dict(tmax=tmax, dx=dx, diffusion_rate=0.05, bcond=bcond, stencil_size_interior=3, stencil_size_boundary=3)
    pde = pnmol.pde.examples.spruce_budworm_1d_discretized(kernel=pnmol.kernels.SquareExponential(),
                                                           nugget_gram_matrix_fd=0.0, **kw)
    opde = oracle.spruce_budworm_1d_discretized(kernel=oracle.SquareExponential(), **kw)
    k = (pnmol.kernels.SquareExponential() if prior == "se" else pnmol.kernels.Matern52()) + pnmol.kernels.WhiteNoise()
    ok = (oracle.SquareExponential() if prior == "se" else oracle.Matern52()) + oracle.WhiteNoise()
    solver = pnmol.white.SemiLinearWhiteNoiseEK1(num_derivatives=nu, spatial_kernel=k,
                                                 steprule=steprule or pnmol.odetools.step.Constant(dt))
    osolver = oracle.WhiteNoiseEK1(num_derivatives=nu, spatial_kernel=ok, semilinear=True, canonical_factor_signs=True,
                                   steprule=osteprule or oracle.Constant(dt))
    return pde, solver, opde, osolver


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_reference_smoke_case(hip_ctx, bcond):
    """tests/test_pdefilter.py:67-94,141-146 (spruce budworm, dx=0.2, tmax=1, Constant(0.1), nu=2, SE+WhiteNoise):
    no NaN -- and here also numerical parity with the oracle on the ten regular steps."""
    pde, solver, opde, osolver = _pair(6, 0.2, 0.1, 1.0, bcond)
    sol = solver.solve(pde)
    osol = osolver.solve(opde)
    assert not np.isnan(sol.mean).any() and not np.isnan(sol.cov_sqrtm).any()
    assert np.array_equal(sol.t, osol.t) and sol.info == osol.info and sol.info["num_steps"] == 11
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(sol.mean[:-1, 0], sol.marginal_std[:-1, 0], omeans[:-1], ostds[:-1])
    np.testing.assert_allclose(sol.mean[:-1], osol.mean[:-1], rtol=1e-5, atol=1e-5 * np.abs(osol.mean).max())


def test_semilinear_larger_mesh(hip_ctx):
    dt, K = 2.0 ** -6, 24
    pde, solver, opde, osolver = _pair(48, 1.0 / 47, dt, K * dt, "dirichlet", prior="matern")
    states = list(solver.solution_generator(pde))
    ostates = list(osolver.solution_generator(opde))
    assert len(states) == len(ostates) == K + 1
    for (s, _), (so, _) in zip(states[1:], ostates[1:]):
        assert s.t == so.t
        np.testing.assert_allclose(s.y.mean, so.y.mean, rtol=1e-5, atol=1e-5 * np.abs(so.y.mean).max())
        ovar = np.einsum("ij,ij->i", so.y.cov_sqrtm, so.y.cov_sqrtm).reshape(so.y.mean.shape, order="F")
        np.testing.assert_allclose(np.sqrt(np.maximum(s.y.marginal_var[0], 0)), np.sqrt(ovar[0]), rtol=1e-4,
                                   atol=1e-5 * np.sqrt(ovar[0]).max())
        np.testing.assert_allclose(s.error_estimate, so.error_estimate, rtol=1e-5)
        np.testing.assert_allclose(s.diffusion_squared_local, so.diffusion_squared_local, rtol=1e-5)
    # the nonlinearity matters: a linear heat solve of the same problem gives a visibly different answer
    lin = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt),
                                          spatial_kernel=solver.spatial_kernel)
    lin_final, _ = lin.simulate_final_state(pde)
    assert np.abs(lin_final.y.mean[0] - states[-1][0].y.mean[0]).max() > 1e-3


def test_adaptive_steps_follow_the_oracle(hip_ctx):
    """Adaptive rule (odetools/step.py:58-119): same accept/reject sequence, time grid and means as the oracle.
    Every rejected attempt re-uses the untouched input state (functional attempt_step)."""
    kw = dict(abstol=1e-3, reltol=1e-2)
    pde, solver, opde, osolver = _pair(12, 1.0 / 11, None, 0.4, "neumann", steprule=pnmol.odetools.step.Adaptive(**kw),
                                       osteprule=oracle.Adaptive(**kw))
    sol = solver.solve(pde)
    osol = osolver.solve(opde)
    assert sol.info == osol.info and sol.info["num_attempted_steps"] >= sol.info["num_steps"] > 3
    np.testing.assert_allclose(sol.t, osol.t, rtol=1e-9)
    np.testing.assert_allclose(sol.mean[:, 0], osol.mean[:, 0], rtol=1e-5, atol=1e-5 * np.abs(osol.mean).max())


def test_diagonal_operator_update_equals_the_dense_one(hip_ctx):
    """`pnmol_filter_set_operator_diagonal` (pointwise nonlinearity: the device patches its stencil rows from diag(J_x))
    against `pnmol_filter_set_operator` with the dense J_x + L: the same ELL image, hence the same numbers, bit for bit."""
    dt, K = 2.0 ** -6, 10
    runs = []
    for diagonal in (True, False):
        pde, solver, _, _ = _pair(48, 1.0 / 47, dt, K * dt, "dirichlet", prior="matern")
        assert pde.df_diagonal is not None
        if not diagonal:
            pde.df_diagonal = None
        runs.append([(s.y.mean.copy(), s.y.marginal_var.copy(), s.error_estimate, s.diffusion_squared_local)
                     for s, _ in list(solver.solution_generator(pde))[1:]])
    for (m1, v1, e1, s1), (m2, v2, e2, s2) in zip(*runs):
        assert np.array_equal(m1, m2) and np.array_equal(v1, v2) and np.array_equal(e1, e2) and s1 == s2
