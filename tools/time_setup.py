"""Where the set-up time of a solve goes (cold path).  Default: the 64x64 2-d mesh of
BASELINE config 5 (D = 8192, m = 4348).  Prints one JSON line of wall-clock seconds per stage."""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
import numpy as np
import pnmol
from pnmol import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt, K = 2.0 ** -9, 2
t = {}
def lap(name, t0):
    t[name] = round(time.perf_counter() - t0, 3)
    return time.perf_counter()

_hip.Context.default(0).synchronize()     # device + library up
t0 = time.perf_counter()
pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(n, n), tmax=K * dt, diffusion_rate=0.05,
                                                       kernel=pnmol.kernels.SquareExponential())
t0 = lap("discretize (mesh, FD weights, dense L/E/B)", t0)
s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                    spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
s.iwp, s.E0, s.E1, gamma = s.initialize_iwp(pde)
t0 = lap("initialize_iwp (Gram, chol, projections)", t0)
s._bind(pde, gamma)
s._device_filter.ctx.synchronize()
t0 = lap("bind (pnmol_filter_create)", t0)
mean, dev = s._initialize_on_device(pde, gamma)
s._device_filter.ctx.synchronize()
t0 = lap("initialize_on_device (two update_sqrt + state upload)", t0)
t["total"] = round(sum(t.values()), 3)
print(json.dumps({"mesh": f"{n}x{n}", "seconds": t}))

# finer: where the two large stages spend their time (second pass, same process: allocations are warm)
if len(sys.argv) > 2 and sys.argv[2] == "--fine":
    import functools
    from pnmol.base import sqrt as dsqrt
    acc = {}
    def timed(name, fn):
        @functools.wraps(fn)
        def w(*a, **k):
            t1 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t1
            return r
        return w
    dsqrt.update_sqrt = timed("update_sqrt (device QR incl. copies)", dsqrt.update_sqrt)
    _hip.Context.cholesky = timed("Context.cholesky (device, incl. copies)", _hip.Context.cholesky)
    _hip.State.set_sqrtm = timed("State.set_sqrtm (upload + C C^T)", _hip.State.set_sqrtm)
    s.spatial_kernel.__class__.__call__ = timed("spatial_kernel(X, X^T) (host)", s.spatial_kernel.__class__.__call__)
    t1 = time.perf_counter()
    s.iwp, s.E0, s.E1, gamma = s.initialize_iwp(pde)
    a = time.perf_counter() - t1
    t1 = time.perf_counter()
    mean, dev = s._initialize_on_device(pde, gamma)
    s._device_filter.ctx.synchronize()
    b = time.perf_counter() - t1
    print(json.dumps({"fine": {k: round(v, 3) for k, v in acc.items()}, "initialize_iwp": round(a, 3),
                      "initialize_on_device": round(b, 3)}))
