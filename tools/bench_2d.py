"""Secondary measurement: BASELINE config 5's mesh (64x64 Dirichlet heat, nu=1, D=8192, m=4348) in fp64 -- ms per step
of the device loop (not a bench.py line: the headline metric is the 1-D N=512 case)."""
import pathlib, sys, time, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
import numpy as np
import pnmol
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"      # "f32": fp32 covariance (pnmol_filter_desc.dtype = 1)
dt, K = 2.0 ** -9, 6
pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                       kernel=pnmol.kernels.SquareExponential())
solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                         spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
solver.dtype = dtype
t0 = time.perf_counter()
state = solver.initialize(pde)
t_init = time.perf_counter() - t0
flt, dev = solver._device_filter, state.y.device_state
solver._ensure_error_model(pde, dt)
flt.steps(dev, 2, dt)
flt.prepare_steps(dev, K, dt)
m, s, infos = flt.steps(dev, K, dt)
ms = flt.last_steps_ms() / K
d = flt.dims()
D, mm, nn = d["n"] * d["d"], d["m"], d["n"]
falg = mm ** 3 / 3 + mm ** 2 * D + D ** 2 * mm + 4 * nn * D ** 2 + 8 * (D * mm + mm ** 2)
print(json.dumps({"mesh": f"{n}x{n}", "dtype": dtype, "D": D, "m": mm, "ms_per_step": ms, "tflops_alg": falg / ms / 1e9,
                  "host_init_s": t_init, "finite": bool(np.isfinite(m).all())}))
