"""Tolerance study of the fp32-covariance mode (pnmol_filter_desc.dtype = 1, `solver.dtype = "f32"`), BASELINE config 5.

Errors of the posterior mean and marginal std, relative to the LARGEST mean / std of the run (the covariance form resolves
a variance to eps * |P-|, so entries much smaller than the largest carry no relative accuracy in any precision):
  * 2-d heat, 12x12 and 28x28 meshes, nu=1: fp64 GPU and fp32 GPU against the CPU oracle (reference algorithm, fp64);
  * 2-d heat, 64x64 mesh (config 5: D=8192, m=4348): fp32 GPU against the fp64 GPU path, and ms/step of both;
  * 1-d heat, N=256 / 512, nu=2: fp32 GPU against the fp64 GPU path.
Prints one JSON line per case."""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np
import pnmol
import pnmol_oracle as oracle


def solve2d(n, K, dt, dtype):
    pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                           kernel=pnmol.kernels.SquareExponential())
    s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                        spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    t, means, stds, sig, _ = s.solve_marginals(pde)
    return means, stds, sig, s._device_filter.last_steps_ms() / K


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def err_rows(tag, m, s, mr, sr):
    # std error also for the entries that matter (>= 1 % of the largest std): the rest is the resolution floor
    big = sr >= 1e-2 * sr.max()
    return {f"{tag}_mean": rel(m, mr), f"{tag}_std": rel(s, sr), f"{tag}_std_rel_on_significant": float(np.abs(s - sr)[big].max() / sr[big].min())}


for n, K in ((12, 12), (28, 8)):
    dt = 2.0 ** -8
    opde = oracle.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05, kernel=oracle.SquareExponential())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    osol = osolver.solve(opde)
    om, os_ = oracle.read_mean_and_std(osol, osolver.E0)
    row = {"case": f"2-d {n}x{n} nu=1, {K} steps, vs CPU oracle"}
    for dtype in ("f64", "f32"):
        m, s, sig, ms = solve2d(n, K, dt, dtype)
        row.update(err_rows(dtype, m, s, om, os_))
        row[f"{dtype}_sigma2_rel"] = float(np.abs(np.mean(sig) - osol.diffusion_squared_calibrated) / osol.diffusion_squared_calibrated)
    print(json.dumps(row), flush=True)

if "--no-big" not in sys.argv:
    K, dt = 4, 2.0 ** -9
    m64, s64, sig64, ms64 = solve2d(64, K, dt, "f64")
    m32, s32, sig32, ms32 = solve2d(64, K, dt, "f32")
    row = {"case": f"2-d 64x64 nu=1 (config 5), {K} steps, fp32 vs fp64 GPU", "ms_per_step_f64": ms64, "ms_per_step_f32": ms32}
    row.update(err_rows("f32", m32, s32, m64, s64))
    row["f32_sigma2_rel"] = float(np.abs(sig32 - sig64).max() / np.abs(sig64).max())
    print(json.dumps(row), flush=True)

sys.path.insert(0, str(ROOT))
import bench
for N, K in ((256, 40), (512, 40)):
    out = {}
    for dtype in ("f64", "f32"):
        pde, solver = bench.build_problem(0.05, K, N)
        solver.dtype = dtype
        solver.allow_unstable_f32 = True          # the point of this case is to show the divergence
        t, means, stds, sig, _ = solver.solve_marginals(pde)
        out[dtype] = (means, stds, sig, solver._device_filter.last_steps_ms() / K)
    row = {"case": f"1-d N={N} nu=2, {K} steps, fp32 vs fp64 GPU", "ms_per_step_f64": out["f64"][3], "ms_per_step_f32": out["f32"][3]}
    row.update(err_rows("f32", out["f32"][0], out["f32"][1], out["f64"][0], out["f64"][1]))
    print(json.dumps(row), flush=True)
