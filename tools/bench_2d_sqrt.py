"""Secondary measurement: the 64x64 2-D mesh (BASELINE config 5's shape, fp64) in square-root (QR) form."""
import pathlib, sys, time, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
import numpy as np
import pnmol
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "f64"      # "f32": the QRs in fp32
dt = 2.0 ** -9
pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                       kernel=pnmol.kernels.SquareExponential())
s = pnmol.sqrtform.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                       spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
s.dtype = DTYPE
t0 = time.perf_counter()
state = s.initialize(pde)
t_init = time.perf_counter() - t0
s._load(state, pde)
flt = s._sqrt_filter
out = []
for k in (1, 1, K):          # as written / + Rc build / steady one-QR steps
    flt.steps(k, dt)
    out.append(flt.last_steps_ms() / k)
m, C = flt.get_state()[1:]
print(json.dumps({"mesh": f"{n}x{n}", "dtype": DTYPE, "D": 2 * n * n, "ms_step_two_qr": out[0], "ms_step_with_rc_build": out[1],
                  "ms_step_one_qr": out[2], "init_s": t_init, "finite": bool(np.isfinite(m).all() and np.isfinite(C).all())}))
