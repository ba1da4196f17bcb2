"""fp32-covariance mode (`pnmol_filter_desc.dtype = 1`, `solver.dtype = "f32"`; BASELINE config 5: "fp32 with tolerance
study").  The covariance and its bulk kernels (predict, stencil gather, down-date on v_mfma_f32_16x16x4_f32) are fp32, the
factorisation of the innovation matrix, the mean and every scalar fp64.  These tests pin what DESIGN.md section 11 claims
for it; tools/fp32_study.py makes the table (incl. the 64x64 mesh, where only the fp64 GPU path can be the yardstick)."""

import numpy as np
import pytest

import pnmol
import pnmol_oracle as oracle
from pnmol import _hip

pytestmark = pytest.mark.gpu


def _solve_2d(n, K, dt, dtype):
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                           kernel=pnmol.kernels.SquareExponential())
    s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                        spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    return s.solve_marginals(pde)


@pytest.mark.parametrize("n,K,floor", [(12, 12, 5e-3), (28, 8, 1.5e-2)])
def test_fp32_covariance_against_the_oracle(hip_ctx, n, K, floor):
    """2-d Dirichlet heat problem, nu=1 (config 5's shape at sizes the oracle can do).  Mean: north_star's rtol 1e-5 holds
    with five digits to spare (the mean update is fp64 and the gain only enters through W).  Std: rtol 1e-4 on every entry
    that is at least 1 % of the largest std; below that the fp32 variance floor eps_32 |P-| takes over (exact-zero
    Dirichlet nodes come out as `floor` x the largest std), which no tolerance on the covariance form can remove."""
    dt = 2.0 ** -8
    opde = oracle.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05, kernel=oracle.SquareExponential())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    osol = osolver.solve(opde)
    om, os_ = oracle.read_mean_and_std(osol, osolver.E0)
    t, means, stds, sig, _ = _solve_2d(n, K, dt, "f32")
    assert np.array_equal(t, osol.t)
    np.testing.assert_allclose(means, om, rtol=1e-5, atol=1e-9 * np.abs(om).max())
    big = os_ >= 1e-2 * os_.max()
    np.testing.assert_allclose(stds[big], os_[big], rtol=1e-4)
    assert np.abs(stds - os_).max() <= floor * os_.max()
    np.testing.assert_allclose(np.mean(sig), osol.diffusion_squared_calibrated, rtol=1e-5)


def test_fp32_state_is_stored_in_single_precision(hip_ctx):
    """set / get through the C ABI stay double; the device copy of the covariance is fp32 (values come back rounded)."""
    n, K, dt = 8, 2, 2.0 ** -8
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
    s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                        spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = "f32"
    s.initialize(pde)
    flt = s._device_filter
    rng = np.random.default_rng(1)
    D = 2 * n * n
    A = rng.standard_normal((D, D))
    cov, mean = A @ A.T, rng.standard_normal((2, n * n))
    st = flt.new_state()
    st.set(0.0, mean, cov)
    np.testing.assert_array_equal(st.mean(), mean)                               # the mean is fp64
    np.testing.assert_array_equal(st.cov(), cov.astype(np.float32).astype(np.float64))
    cl = st.clone()
    np.testing.assert_array_equal(cl.cov(), st.cov())


def test_fp32_needs_the_fused_sweep(hip_ctx):
    """nu = 3 (n = 4) runs the stand-alone down-date kernel, which is fp64 only: the mode is refused, not silently widened."""
    pde = pnmol.pde.examples.heat_1d_discretized(tmax=0.1, dx=1.0 / 15, kernel=pnmol.kernels.SquareExponential())
    s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=3, steprule=pnmol.odetools.step.Constant(0.01),
                                        spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = "f32"
    with pytest.raises(ValueError, match="num_derivatives = 1"):          # the host mirror: nu >= 2 diverges in fp32
        s.initialize(pde)
    s.allow_unstable_f32 = True
    with pytest.raises(_hip.PnmolHipError):                               # the library: no fp32 down-date kernel for nu = 3
        s.initialize(pde)
