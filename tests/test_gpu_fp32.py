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


def _floor_model(n, K):
    """tools/fp32_floor_model.py: the fp64 covariance-form step of the oracle with exactly the device's fp32 roundings."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location(
        "fp32_floor_model", pathlib.Path(__file__).resolve().parents[1] / "tools" / "fp32_floor_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.model(n, K)


@pytest.mark.parametrize("n,K", [(12, 12), (28, 8)])
def test_fp32_covariance_against_the_oracle(hip_ctx, n, K):
    """2-d Dirichlet heat problem, nu=1 (config 5's shape at sizes the oracle can do).  Mean: north_star's rtol 1e-5 holds
    with five digits to spare (the mean update is fp64 and the gain only enters through W).  Std: rtol 1e-4 on every entry
    that is at least 1 % of the largest std.  Below that the mode has a floor, and the floor is not an asserted constant:
    it is PREDICTED by a CPU model that applies the device's three fp32 roundings (P, P-, the fp32-accumulated down-date) to
    the oracle's fp64 covariance-form step (tools/fp32_floor_model.py; the same model shows that an fp64 copy of diag(P)
    beside the fp32 matrix -- VERDICT round 2, item 1b -- makes the floor 10x worse, which is why it is not built:
    profiles/r03_fp32_floor_model.log).  The device may not exceed the model by more than 1.5x."""
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
    predicted = _floor_model(n, K)
    assert predicted["rel_on_significant_fp32_matrix"] < 1e-4          # the model agrees that the significant entries hold
    assert predicted["floor_fp64_shadow"] > 3 * predicted["floor_fp32_matrix"]   # ... and that the shadow would not help
    floor = np.abs(stds - os_).max() / os_.max()
    assert floor <= 1.5 * predicted["floor_fp32_matrix"], (floor, predicted)
    np.testing.assert_allclose(np.mean(sig), osol.diffusion_squared_calibrated, rtol=1e-5)


def test_fp32_at_config5_size_against_the_fp64_device_path(hip_ctx):
    """BASELINE config 5 as stated: 64x64 mesh, nu = 1 (D = 8192, m = 4348), fp32 covariance, 4 steps.  The oracle needs
    minutes per step here, so the yardstick is the fp64 device path (itself compared with the oracle at 12x12 / 28x28 above
    and property-checked at this size in test_gpu_parity.py): mean 1e-5, std 1e-4 on the significant entries, the floor within
    the study's figure (1.7e-2 of max(std); the CPU model gives 1.3e-3 / 2.2e-3 / 5.3e-3 at 12 / 20 / 28
    points per side, growing with 1/dx^2 as the stencil amplifies the fp32 rounding of P-), plus the size-independent
    properties: finite, boundary nodes pinned, non-negative variances, bit-reproducible."""
    dt, K = 2.0 ** -9, 4
    t64, m64, s64, sig64, _ = _solve_2d(64, K, dt, "f64")
    t32, m32, s32, sig32, fin32 = _solve_2d(64, K, dt, "f32")
    assert np.array_equal(t32, t64) and m32.shape == (K + 1, 4096)
    assert np.all(np.isfinite(m32)) and np.all(np.isfinite(s32)) and np.all(np.isfinite(sig32)) and np.all(sig32 > 0)
    np.testing.assert_allclose(m32, m64, rtol=1e-5, atol=1e-9 * np.abs(m64).max())
    big = s64 >= 1e-2 * s64.max()
    np.testing.assert_allclose(s32[big], s64[big], rtol=1e-4)
    assert np.abs(s32 - s64).max() <= 2.5e-2 * s64.max()
    np.testing.assert_allclose(sig32, sig64, rtol=1e-4)
    var = fin32.y.marginal_var
    assert var.min() > -1e-6 * var.max()
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(64, 64), tmax=K * dt, diffusion_rate=0.05,
                                                           kernel=pnmol.kernels.SquareExponential())
    on_boundary = pde.mesh_spatial.boundary[1]
    assert np.abs(m32[-1][on_boundary]).max() < 1e-8
    t32b, m32b, s32b, sig32b, _ = _solve_2d(64, K, dt, "f32")
    assert np.array_equal(m32, m32b) and np.array_equal(s32, s32b) and np.array_equal(sig32, sig32b)


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
