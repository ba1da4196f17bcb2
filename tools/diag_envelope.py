"""Probes beyond the sizes the GPU suite covers (finite? agreement between independent device paths?).  One JSON line per case."""
import sys, pathlib, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np, pnmol
DT = 2.0 ** -7


def pde1d(N, K):
    return pnmol.pde.examples.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3, stencil_size_boundary=3,
                                                  t0=0.0, tmax=K * DT, diffusion_rate=0.05, kernel=pnmol.kernels.SquareExponential(),
                                                  nugget_gram_matrix_fd=0.0, bcond="dirichlet")


def solver(mod, nu, dtype="f64"):
    s = mod.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(DT),
                                spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    s.dtype = dtype
    return s


def cmp(tag, a, b):
    (ma, sa), (mb, sb) = a, b
    fin = bool(np.isfinite(ma).all() and np.isfinite(sa).all() and np.isfinite(mb).all() and np.isfinite(sb).all())
    row = {"case": tag, "finite": fin}
    if fin:
        big = sb > 1e-2 * sb.max()
        row.update(mean_rel=float(np.abs(ma - mb).max() / np.abs(mb).max()), std_floor=float(np.abs(sa - sb).max() / sb.max()),
                   std_rel_significant=float((np.abs(sa - sb)[big] / sb[big]).max()))
    print(json.dumps(row), flush=True)


def run(s, pde):
    t, m, sd, sig, _ = s.solve_marginals(pde)
    return m, sd


# 1. covariance form against square-root form, both fp64, at config 3's size
p = pde1d(1024, 12)
cmp("N=1024 nu=2, 12 steps: covariance form vs square-root form (fp64)", run(solver(pnmol.white, 2), p), run(solver(pnmol.sqrtform, 2), p))
# 2. fp32 covariance mode where it is allowed (nu = 1) in 1-d at sizes beyond the 2-d tests
for N in (512, 1024):
    p = pde1d(N, 20)
    cmp(f"N={N} nu=1, 20 steps: fp32 covariance vs fp64 covariance", run(solver(pnmol.white, 1, "f32"), p), run(solver(pnmol.white, 1), p))
# 3. a size between the register-resident sweep (CB <= 17) and config 3
p = pde1d(640, 12)
cmp("N=640 nu=2, 12 steps: covariance form vs square-root form (fp64)", run(solver(pnmol.white, 2), p), run(solver(pnmol.sqrtform, 2), p))
# 4. nu = 3 at N = 512 (k_sweep + stand-alone down-date)
p = pde1d(512, 8)
cmp("N=512 nu=3, 8 steps: covariance form vs square-root form (fp64)", run(solver(pnmol.white, 3), p), run(solver(pnmol.sqrtform, 3), p))
