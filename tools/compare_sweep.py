"""One eager filter step from the same initial state; dumps F = [Ls; W; r; Ls^-T], the posterior mean/covariance.
Run twice (PNMOL_HIP_SWEEP_RL=0 / 1) and compare with --diff."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np

if sys.argv[1] == "--diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    mp = int(a["mp"])
    for k in ("F", "mean", "cov"):
        d = np.abs(a[k] - b[k])
        print(k, "max abs diff", d.max(), "rel to max", d.max() / np.abs(a[k]).max())
    d = np.abs(a["F"] - b["F"])
    nb = d.shape[0] // 32
    tiles = d.reshape(nb, 32, mp // 32, 32).max(axis=(1, 3))
    ref = np.abs(a["F"]).reshape(nb, 32, mp // 32, 32).max(axis=(1, 3))
    np.set_printoptions(linewidth=250, precision=1)
    print("per-tile max abs diff (rows = 32-row blocks of the tall matrix):")
    print(tiles)
    print("per-tile max |F|:")
    print(ref)
    sys.exit(0)

from helpers import make_pair
N, out = int(sys.argv[1]), sys.argv[2]
dt = 2.0 ** -7
pde, solver, opde, osolver = make_pair(N, 2, dt, 8)
s0 = solver.initialize(pde)
flt = solver._device_filter
solver._ensure_error_model(pde, dt)
dev = flt.new_state()
dev.set(0.0, s0.y.mean, s0.y.cov)
cur = dev
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    cur, info, err = flt.step(cur, dt)
dims = flt.dims()
mp, dp = dims["mp"], dims["dp"]
rows = mp + 3 * dp + 32 + mp
F = flt.debug_read(2, rows * mp).reshape(rows, mp)
np.savez(out, F=F, mean=cur.mean(), cov=cur.cov(), mp=mp)
print("saved", out, "info", info.info, info.diffusion_squared_local)
