"""Tolerance study of the fp32 QR mode of the square-root form (`pnmol.sqrtform.*` with `dtype = "f32"`,
`pnmol_filter_desc.dtype = 1` on `pnmol_sqrt_filter_create`) -- the companion of tools/fp32_study.py, same error measures:
errors of the posterior mean / marginal std relative to the LARGEST mean / std ("floor"), and the relative std error on the
entries >= 1 % of the largest std ("significant").
  * 1-d heat, nu = 2, N = 64 / 128 / 256 (40 / 40 / 100 steps): fp32 QR and fp64 QR on the GPU against the CPU oracle
    (N = 256: the committed 100-step fixture of BASELINE config 2) -- the workload on which the fp32 COVARIANCE form diverges;
  * 1-d heat, nu = 2, N = 512 (30 steps) and N = 1024 (24 steps): fp32 QR against the fp64 QR form on the GPU, ms per step of both;
  * 2-d heat, nu = 1, 12x12 and 28x28: against the CPU oracle (the fp32 covariance form's floor there: 1.3e-3 / 4.1e-3).
Prints one JSON line per case (log: profiles/r03_sqrt_fp32/fp32_sqrt_study.log)."""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np
import pnmol
import pnmol_oracle as oracle

DT = 2.0 ** -7


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def err_rows(tag, m, s, mr, sr):
    big = sr >= 1e-2 * sr.max()
    return {f"{tag}_mean": rel(m, mr), f"{tag}_std_floor": rel(s, sr),
            f"{tag}_std_rel_on_significant": float((np.abs(s - sr)[big] / sr[big]).max())}


def sqrt_solver(nu, dt, dtype):
    s = pnmol.sqrtform.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
                                           spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    return s


def heat1d(mod, N, K):
    return mod.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3, stencil_size_boundary=3, t0=0.0,
                                   tmax=K * DT, diffusion_rate=0.05, kernel=(mod if mod is oracle else pnmol.kernels).SquareExponential(),
                                   nugget_gram_matrix_fd=0.0, bcond="dirichlet")


for N, K in ((64, 40), (128, 40), (256, 100)):
    if N == 256:
        f = np.load(ROOT / "tests" / "golden" / "oracle_heat_n256_nu2_kc.npz")
        om, os_ = f["means"], f["stds"]
    else:
        osolver = oracle.WhiteNoiseEK1(num_derivatives=2, steprule=oracle.Constant(DT), canonical_factor_signs=True,
                                       spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
        om, os_ = oracle.read_mean_and_std(osolver.solve(heat1d(oracle, N, K)), osolver.E0)
    row = {"case": f"1-d N={N} nu=2, {K} steps, QR form vs CPU oracle"}
    for dtype in ("f64", "f32"):
        s = sqrt_solver(2, DT, dtype)
        t, m, sd, sig, _ = s.solve_marginals(heat1d(pnmol.pde.examples, N, K))
        row.update(err_rows(dtype, m, sd, om, os_))
    print(json.dumps(row), flush=True)

for N, K in ((512, 30), (1024, 24)):
    out = {}
    for dtype in ("f64", "f32"):
        s = sqrt_solver(2, DT, dtype)
        t, m, sd, sig, _ = s.solve_marginals(heat1d(pnmol.pde.examples, N, K))
        out[dtype] = (m, sd, s._sqrt_filter.last_steps_ms() / K)
        del s
    row = {"case": f"1-d N={N} nu=2, {K} steps, fp32 QR vs fp64 QR on the GPU", "ms_per_step_f64": out["f64"][2], "ms_per_step_f32": out["f32"][2]}
    row.update(err_rows("f32", out["f32"][0], out["f32"][1], out["f64"][0], out["f64"][1]))
    print(json.dumps(row), flush=True)

for n, K in ((12, 12), (28, 8)):
    dt = 2.0 ** -8
    opde = oracle.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05, kernel=oracle.SquareExponential())
    osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
    om, os_ = oracle.read_mean_and_std(osolver.solve(opde), osolver.E0)
    row = {"case": f"2-d {n}x{n} nu=1, {K} steps, QR form vs CPU oracle"}
    for dtype in ("f64", "f32"):
        pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                               kernel=pnmol.kernels.SquareExponential())
        t, m, sd, sig, _ = sqrt_solver(1, dt, dtype).solve_marginals(pde)
        row.update(err_rows(dtype, m, sd, om, os_))
    print(json.dumps(row), flush=True)
