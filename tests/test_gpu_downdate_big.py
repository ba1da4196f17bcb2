"""The large-problem down-date (`k_downdate_big`: P = P- - W W^T as one SYRK on 128x128 tiles, used from D = 8192 on)
against the default path on a problem small enough for both, and against the oracle.  Reference arithmetic: the
`P- - K S K^T` term of `update_sqrt` in covariance form (src/pnmol/base/sqrt.py:33-73, white.py:120-135)."""
import os

import numpy as np
import pytest

import pnmol_oracle as oracle
from helpers import assert_mean_std_parity

pytestmark = pytest.mark.gpu


def _solve(nums, K, dt, force, dtype="f64"):
    import pnmol

    old = os.environ.get("PNMOL_HIP_DD_BIG")
    os.environ["PNMOL_HIP_DD_BIG"] = force          # read when the filter is created
    try:
        pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=nums, tmax=K * dt, diffusion_rate=0.05,
                                                               kernel=pnmol.kernels.SquareExponential())
        solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                                 spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
        solver.dtype = dtype
        t, means, stds, sig, final = solver.solve_marginals(pde)
        cov = np.array(final.y.cov)
        return means, stds, sig, cov
    finally:
        if old is None:
            del os.environ["PNMOL_HIP_DD_BIG"]
        else:
            os.environ["PNMOL_HIP_DD_BIG"] = old


@pytest.mark.parametrize("nums", [(16, 16), (24, 16)])
def test_big_downdate_equals_the_default_path(hip_ctx, nums):
    """D = 2 * 256 = 512 (4 x 4 tiles of 128, diagonal and off-diagonal, mirror images) and D = 768; m = 316 / 456 is not a
    multiple of 128.  Same W, different association order of the m products per entry: agreement to rounding."""
    dt, K = 2.0 ** -9, 6
    m0, s0, g0, c0 = _solve(nums, K, dt, "0")
    m1, s1, g1, c1 = _solve(nums, K, dt, "1")
    np.testing.assert_allclose(m1, m0, rtol=0, atol=1e-13 * np.abs(m0).max())
    # (a variance is resolved to eps |P-|, a std to its root: entries that are exactly 0 come out as 1e-11 .. 1e-10 either way)
    np.testing.assert_allclose(s1, s0, rtol=1e-7, atol=1e-6 * s0.max())
    np.testing.assert_allclose(g1, g0, rtol=1e-9)
    np.testing.assert_allclose(c1, c0, rtol=0, atol=1e-13 * np.abs(c0).max())
    np.testing.assert_allclose(c1, c1.T, rtol=0, atol=1e-13 * np.abs(c0).max())


def test_big_downdate_against_the_oracle(hip_ctx):
    dt, K = 2.0 ** -9, 4
    means, stds, sig, cov = _solve((16, 16), K, dt, "1")
    opde = oracle.heat_2d_dirichlet_discretized(nums=(16, 16), tmax=K * dt, diffusion_rate=0.05,
                                                kernel=oracle.SquareExponential())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    osol = osolver.solve(opde)
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(means, stds, omeans, ostds)


def test_big_downdate_fp32_covariance(hip_ctx):
    """dtype = "f32" (BASELINE config 5's precision): the fp32 SYRK against the fp32 form of the fused pairs -- the same
    arithmetic types (W rounded to fp32, v_mfma_f32_16x16x4_f32, fp32 accumulators) in another association order: agreement
    at the resolution of an fp32 covariance (6e-8 |P|), i.e. the floor DESIGN.md section 11 describes for the stds."""
    dt, K = 2.0 ** -9, 4
    m0, s0, g0, c0 = _solve((16, 16), K, dt, "0", "f32")
    m1, s1, g1, c1 = _solve((16, 16), K, dt, "1", "f32")
    np.testing.assert_allclose(m1, m0, rtol=0, atol=1e-7 * np.abs(m0).max())
    np.testing.assert_allclose(c1, c0, rtol=0, atol=5e-6 * np.abs(c0).max())
    big = s0 > 1e-2 * s0.max()
    np.testing.assert_allclose(s1[big], s0[big], rtol=1e-3)
    np.testing.assert_allclose(g1, g0, rtol=1e-4)


def test_w_as_a_gemm_behind_the_sweep(hip_ctx):
    """32 x 20 mesh: D = 1280, m = 740 -> 768 = 6 x 128 padded columns, 24 column blocks: the large-problem path with the rows of
    W taken out of the sweep (`k_w_gemm`: W = (P- H^T) Ls^-T with the Ls^-T the sweep leaves in the identity rows) against the
    fused launch, and against the oracle.  Same operator applied to the same rows, as one product instead of a forward
    substitution: agreement to rounding times the conditioning of S."""
    dt, K = 2.0 ** -9, 3
    m0, s0, g0, c0 = _solve((32, 20), K, dt, "0")
    m1, s1, g1, c1 = _solve((32, 20), K, dt, "1")
    np.testing.assert_allclose(m1, m0, rtol=0, atol=1e-11 * np.abs(m0).max())
    np.testing.assert_allclose(c1, c0, rtol=0, atol=1e-11 * np.abs(c0).max())
    np.testing.assert_allclose(s1, s0, rtol=1e-6, atol=1e-5 * s0.max())
    np.testing.assert_allclose(g1, g0, rtol=1e-8)
    opde = oracle.heat_2d_dirichlet_discretized(nums=(32, 20), tmax=K * dt, diffusion_rate=0.05,
                                                kernel=oracle.SquareExponential())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    osol = osolver.solve(opde)
    omeans, ostds = oracle.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(m1, s1, omeans, ostds)
