"""Secondary measurement: the square-root (QR) form of the step (pnmol.sqrtform / include/pnmol_sqrt.h) on the headline
problem (1-D heat, N=512, nu=2), device loop, next to the covariance form's step and the CPU oracle's as-written step."""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "pnmol-experiments_amd"), str(ROOT / "oracle")]
import numpy as np
import pnmol
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "f64"      # "f32": the QRs in fp32 (pnmol_filter_desc.dtype = 1)
nu, dt = 2, 2.0 ** -7
kw = dict(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=0.05, bcond="dirichlet", stencil_size_interior=3,
          stencil_size_boundary=3, nugget_gram_matrix_fd=0.0, kernel=pnmol.kernels.SquareExponential())
pde = pnmol.pde.examples.heat_1d_discretized(**kw)
prior = pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise()
s = pnmol.sqrtform.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=prior)
s.dtype = DTYPE
t0 = time.perf_counter()
t, means, stds, sig, final = s.solve_marginals(pde)
wall = time.perf_counter() - t0
ms = s._sqrt_filter.last_steps_ms() / K
c = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=prior)
tc, mc, sc, sigc, _ = c.solve_marginals(pde)
D, m = (nu + 1) * N, N + 2
flops_qr = (2.0 * (2 * D) * D ** 2 - 2.0 / 3 * D ** 3) + (2.0 * (D + m) * (D + m) ** 2 - 2.0 / 3 * (D + m) ** 3)
print(json.dumps({"N": N, "steps": K, "dtype": DTYPE, "sqrt_ms_per_step": ms, "sqrt_qr_tflops": flops_qr / ms / 1e9,
                  "wall_s_incl_init": wall, "cov_ms_per_step": c._device_filter.last_steps_ms() / K,
                  "max_rel_mean_diff": float(np.max(np.abs(means - mc)) / np.abs(mc).max()),
                  "max_rel_std_diff": float(np.max(np.abs(stds[1:] - sc[1:]) / sc[1:].max()))}))
