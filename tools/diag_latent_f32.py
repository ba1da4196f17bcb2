"""fp32 QR mode on the latent-force model (noise-free update, nuggets 1e-6, conditioning ~1e10): fp32 QR against fp64 QR on the GPU."""
import sys, pathlib, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np, pnmol
nu, dt = 2, 2.0 ** -6
for N, K, bcond in ((24, 6, "neumann"), (24, 6, "dirichlet"), (96, 6, "neumann")):
    kw = dict(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=0.05, bcond=bcond, kernel=pnmol.kernels.SquareExponential())
    pde = pnmol.pde.examples.heat_1d_discretized(**kw)
    k = pnmol.kernels.SquareExponential() + pnmol.kernels.WhiteNoise()
    out = {}
    for dtype in ("f64", "f32"):
        s = pnmol.sqrtform.LinearLatentForceEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
        s.dtype = dtype
        t, means, stds, sig, final = s.solve_marginals(pde)
        out[dtype] = (means, stds)
    m64, s64 = out["f64"]; m32, s32 = out["f32"]
    fin = bool(np.isfinite(m32).all() and np.isfinite(s32).all())
    row = {"case": f"latent-force N={N} {bcond} nu=2, {K} steps, fp32 QR vs fp64 QR", "finite": fin}
    if fin:
        big = s64 > 1e-2 * s64.max()
        row.update(mean_rel=float(np.abs(m32 - m64).max() / np.abs(m64).max()), std_floor=float(np.abs(s32 - s64).max() / s64.max()),
                   std_rel_significant=float((np.abs(s32 - s64)[big] / s64[big]).max()))
    print(json.dumps(row), flush=True)
