"""fp32 build of the square-root (QR) form (`pnmol.sqrtform.*` with `dtype = "f32"`, `pnmol_filter_desc.dtype = 1` on
`pnmol_sqrt_filter_create`): the QRs of white.py:114 / :120 (base/sqrt.py:8-95) in fp32 on the device, everything else fp64.

This is the fp32 mode for num_derivatives >= 2, where the fp32 COVARIANCE form diverges (finding 3):
the north-star tolerances (mean 1e-5, std 1e-4) are asserted against the fp64 CPU oracle -- on seeded problems, and at
BASELINE config 2's size and length (N=256, nu=2, 100 steps) against the committed oracle fixture -- with NO std floor
beyond the one the fp64 tests use.  `tools/fp32_sqrt_model.py` is the CPU model that predicted these numbers.
"""

import pathlib

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from helpers import assert_mean_std_parity, make_pair

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).parent / "golden"


def _solver(nu, dt, dtype, semilinear=False, kernel=None, steprule=None):
    cls = pnmol.sqrtform.SemiLinearWhiteNoiseEK1 if semilinear else pnmol.sqrtform.LinearWhiteNoiseEK1
    s = cls(num_derivatives=nu, steprule=steprule or pnmol.odetools.step.Constant(dt),
            spatial_kernel=kernel or pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    return s


@pytest.mark.parametrize("N,nu,bcond,K", [(40, 2, "dirichlet", 10), (64, 2, "neumann", 12), (50, 3, "neumann", 6),
                                          (128, 2, "dirichlet", 8), (33, 1, "dirichlet", 8)])
def test_fp32_qr_solve_matches_oracle(N, nu, bcond, K):
    """Device loop (one QR per step from the second step on) and the step-by-step solve (two QRs per step), fp32 QR,
    against the fp64 oracle at the north-star tolerances."""
    dt = 2.0 ** -7
    pde, _, opde, osolver = make_pair(N, nu, dt, K, bcond=bcond)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    t, means, stds, sig, final = _solver(nu, dt, "f32").solve_marginals(pde)
    assert_mean_std_parity(means, stds, om, os_)
    np.testing.assert_allclose(np.mean(sig), osol.diffusion_squared_calibrated, rtol=2e-3)
    C = final.y.cov_sqrtm
    assert np.all(np.triu(C, 1) == 0) and np.all(np.diag(C) >= 0) and np.all(np.isfinite(C))
    if N <= 64:
        sol = _solver(nu, dt, "f32").solve(pde)
        assert_mean_std_parity(sol.mean[:, 0], sol.marginal_std[:, 0], om, os_)


def test_fp32_qr_really_runs_in_fp32():
    """The mode is not a silent fp64 run: its results differ from the fp64 square-root form at the fp32 level
    (1e-9 .. 1e-4 relative), and the fp64 form is unchanged by the build (1e-10 against the oracle-level solve)."""
    N, nu, dt, K = 64, 2, 2.0 ** -7, 6
    pde, _, _, _ = make_pair(N, nu, dt, K)
    t, m32, s32, _, _ = _solver(nu, dt, "f32").solve_marginals(pde)
    t, m64, s64, _, _ = _solver(nu, dt, "f64").solve_marginals(pde)
    rel = np.abs(s32[1:] - s64[1:]).max() / s64.max()
    assert 1e-9 < rel < 1e-4, rel
    assert np.abs(m32 - m64).max() / np.abs(m64).max() < 1e-5


def test_fp32_qr_full_length_n256_config_2():
    """Where the fp32 covariance form diverges (mean errors of 1e10 after 40 steps): 1-d, nu = 2,
    N = 256, all 100 steps of BASELINE config 2 against the committed fp64 oracle fixture, mean 1e-5 / std 1e-4."""
    f = np.load(GOLD / "oracle_heat_n256_nu2_kc.npz")
    N, dt = 256, 2.0 ** -7
    pde = pnmol.pde.examples.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3,
                                                 stencil_size_boundary=3, t0=0.0, tmax=100 * dt, diffusion_rate=float(f["kappa"]),
                                                 kernel=pnmol.kernels.SquareExponential(), nugget_gram_matrix_fd=0.0,
                                                 bcond="dirichlet")
    t, means, stds, sig, _ = _solver(2, dt, "f32").solve_marginals(pde)
    assert np.array_equal(t, f["t"])
    assert_mean_std_parity(means, stds, f["means"], f["stds"])
    # stds with no floor where they are not exactly zero: relative 1e-4 down to 1e-6 of the largest std
    big = f["stds"] > 1e-6 * f["stds"].max()
    np.testing.assert_allclose(stds[big], f["stds"][big], rtol=1e-4)


def test_fp32_covariance_form_refuses_what_this_mode_covers():
    pde, solver, _, _ = make_pair(32, 2, 2.0 ** -7, 2)
    solver.dtype = "f32"
    with pytest.raises(ValueError, match="sqrtform"):
        solver.solve_marginals(pde)


@pytest.mark.parametrize("semilinear", [False, True])
def test_fp32_qr_adaptive_steps_follow_the_oracle(semilinear):
    """The Adaptive rule on the fp32 QR form (estimate_error's factor in fp32 too; semilinear: two QRs per step, operator
    replaced every step): the oracle's accept/reject sequence, its step sizes to fp32 level, means at 1e-5."""
    kw = dict(abstol=1e-3, reltol=1e-2)
    k, ok = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise(), o.SquareExponential() + o.WhiteNoise()
    pkw = dict(tmax=0.4, dx=1.0 / 11, diffusion_rate=0.05, bcond="neumann")
    if semilinear:
        pde = pnmol.pde.examples.spruce_budworm_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **pkw)
        opde = o.spruce_budworm_1d_discretized(kernel=o.SquareExponential(), **pkw)
    else:
        pde = pnmol.pde.examples.heat_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **pkw)
        opde = o.heat_1d_discretized(kernel=o.SquareExponential(), **pkw)
    solver = _solver(2, None, "f32", semilinear=semilinear, kernel=k, steprule=pnmol.odetools.step.Adaptive(**kw))
    osolver = o.WhiteNoiseEK1(num_derivatives=2, spatial_kernel=ok, semilinear=semilinear, canonical_factor_signs=True,
                              steprule=o.Adaptive(**kw))
    sol, osol = solver.solve(pde), osolver.solve(opde)
    assert sol.info == osol.info and sol.info["num_attempted_steps"] >= sol.info["num_steps"] > 3
    np.testing.assert_allclose(sol.t, osol.t, rtol=1e-4)
    np.testing.assert_allclose(sol.mean[:, 0], osol.mean[:, 0], rtol=1e-5, atol=1e-5 * np.abs(osol.mean[:, 0]).max())


def test_fp32_qr_two_dimensional_mesh():
    """2-d Dirichlet heat problem, nu = 1 (the shape of BASELINE config 5), 12 x 12 against the oracle: the fp32 QR form has
    no std floor there either (the fp32 covariance form: 1.3e-3 of max(std), tests/test_gpu_fp32.py)."""
    dt, K = 2.0 ** -8, 6
    k = pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise()
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
    opde = o.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=o.SquareExponential())
    osolver = o.WhiteNoiseEK1(num_derivatives=1, steprule=o.Constant(dt), canonical_factor_signs=True,
                              spatial_kernel=o.Matern52() + o.WhiteNoise())
    t, means, stds, sig, _ = _solver(1, dt, "f32", kernel=k).solve_marginals(pde)
    osol = osolver.solve(opde)
    om, os_ = o.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(means, stds, om, os_)
