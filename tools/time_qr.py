"""Device time of the QR (R factor) at the sizes of one square-root filter step (N=512, nu=2: predict 3072 x 1536,
update 2050 x 2050) and at LAPACK's on the host cores beside it."""
import pathlib, sys, time, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
import numpy as np, scipy.linalg
from pnmol import _hip
ctx = _hip.Context.default()
rng = np.random.default_rng(0)
for rows, cols in [(3072, 1536), (2050, 2050), (1536, 768), (6144, 3072)]:
    A = rng.standard_normal((rows, cols))
    ctx.qr_r(A)
    ms = []
    for _ in range(3):
        ctx.qr_r(A)
        ms.append(ctx.qr_last_ms())
    t0 = time.perf_counter(); scipy.linalg.qr(A, mode="r"); t_cpu = time.perf_counter() - t0
    flops = 2.0 * rows * cols ** 2 - 2.0 / 3.0 * cols ** 3
    print(json.dumps({"rows": rows, "cols": cols, "gpu_ms": min(ms), "tflops": flops / min(ms) / 1e9,
                      "lapack_ms": t_cpu * 1e3}), flush=True)
