import sys, pathlib, json
ROOT = pathlib.Path("/root/repo")
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests", ROOT / "tools"):
    sys.path.insert(0, str(p))
import numpy as np, pnmol
DT = 2.0 ** -7
def run(N, K, dtype):
    pde = pnmol.pde.examples.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3, stencil_size_boundary=3, t0=0.0,
        tmax=K * DT, diffusion_rate=0.05, kernel=pnmol.kernels.SquareExponential(), nugget_gram_matrix_fd=0.0, bcond="dirichlet")
    s = pnmol.sqrtform.LinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(DT),
                                           spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    t, m, sd, sig, _ = s.solve_marginals(pde)
    bad_m = [int(i) for i in np.where(~np.isfinite(m).all(axis=1))[0][:3]]
    bad_s = [int(i) for i in np.where(~np.isfinite(sd).all(axis=1))[0][:3]]
    return m, sd, bad_m, bad_s
N = int(sys.argv[1]); K = int(sys.argv[2])
out = {}
for dt in ("f64", "f32"):
    m, sd, bm, bs = run(N, K, dt)
    out[dt] = (m, sd)
    print(dt, "first non-finite mean rows", bm, "std rows", bs, flush=True)
if np.isfinite(out["f32"][1]).all() and np.isfinite(out["f64"][1]).all():
    print("std rel", float(np.abs(out["f32"][1] - out["f64"][1]).max() / out["f64"][1].max()))
