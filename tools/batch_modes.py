"""A/B of the launch modes of the step with several problems in flight on one GPU (north_star's kappa sweep).

Usage: [PNMOL_HIP_SWEEP=1] [PNMOL_HIP_SWEEP_RL=0] python tools/batch_modes.py [B] [K]
Prints steps/s of B problems run one after the other and all in flight (one stream each)."""
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pnmol import batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
probs = [bench._bound_problem(batch.diffusion_sweep(g % 8, 8), 2 * K + 4, 0) for g in range(B)]
for _, flt, dev in probs:
    flt.steps(dev, 2, bench.DT)
    flt.prepare_steps(dev, K, bench.DT)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("PNMOL_HIP")}, "problems": B, "steps_each": K}
for mode in ("serial", "concurrent", "concurrent"):
    for _, flt, dev in probs:
        flt.prepare_steps(dev, K, bench.DT)
    for ctx, _, _ in probs:
        ctx.synchronize()
    t0 = time.perf_counter()
    if mode == "serial":
        for _, flt, dev in probs:
            flt.steps_begin(dev, K, bench.DT)
            flt.steps_end(dev, want_means=True, want_stds=True)
    else:
        for _, flt, dev in probs:
            flt.steps_begin(dev, K, bench.DT)
        res = [flt.steps_end(dev, want_means=True, want_stds=True) for _, flt, dev in probs]
        assert all(np.all(np.isfinite(m)) and all(o.info == -1 for o in infos) for m, _, infos in res)
    out.setdefault(mode + "_steps_per_s", []).append(round(B * K / (time.perf_counter() - t0), 1))
print(json.dumps(out))
