import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
import numpy as np
from pnmol import _hip
ctx = _hip.Context.default()
A = np.random.default_rng(0).standard_normal((2050, 2050))
for _ in range(3):
    ctx.qr_r(A)
    print(ctx.qr_last_ms())
