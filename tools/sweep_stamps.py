"""Timeline of one sweep launch (library built with -DPNMOL_SWEEP_STAMP, loaded through PNMOL_HIP_LIB).
k_sweep_rl layout: block 0 = chain workgroup (per-block trace in rows 384 + J), block 1 + I = row block I."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))
sys.path.insert(0, str(ROOT))
import numpy as np
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bench.MESH_N = N
pde, solver = bench.build_problem(0.05, 40)
state = solver.initialize(pde)
flt, dev = solver._device_filter, state.y.device_state
solver._ensure_error_model(pde, bench.DT)
flt.steps(dev, 20, bench.DT)
import os
LOOP = os.environ.get("PNMOL_STAMP_LOOP") == "1"   # stamps of the LAST step of the loop above (folded loop step: front / tail roles)
if not LOOP:
    out, info, err = flt.step(dev, bench.DT)     # one eager step: the stamps of its sweep
st = flt.debug_read(5, 512 * 8).reshape(512, 8)
mp, dp = flt.dims()["mp"], flt.dims()["dp"]
CB = mp // 32
RT = 2 * CB + 3 * dp // 32 + 1
t0 = st[1:256, 0][st[1:256, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
print("chain workgroup, per diagonal block J (us): start | factor (w3 done) | +inverse stored (w1) | +barrier | +TRSM,barrier | +SYRK -> next start")
prev = None
for J in range(CB):
    a = us(st[384 + J])
    nxt = us(st[384 + J + 1])[0] if J + 1 < CB else float("nan")
    print(f"J={J:2d} start {a[0]:7.2f} | factor {a[1]-a[0]:5.2f} | inv {a[6]-a[1]:5.2f} | barrier {a[2]-max(a[1],a[6]):5.2f} (at {a[2]:7.2f}) | trsm {a[3]-a[2]:5.2f} | syrk {nxt-a[3]:5.2f} | period {nxt-a[0]:5.2f}")
clk = st[384:384 + CB, 4].astype(float)
wall = st[384:384 + CB, 0].astype(float)
print("shader clock of the chain workgroup's CU between diagonal blocks (MHz):",
      " ".join(f"{(clk[i+1]-clk[i])/(wall[i+1]-wall[i])*100:.0f}" for i in range(CB - 1)))
rows = us(st[1:1 + RT])
chain_feed = rows[1:CB, 1]
print("chain rows: fed at", " ".join(f"{x:.1f}" for x in chain_feed))
bulk = rows[CB:RT]
print(f"other row blocks {RT-CB}: start min/max {bulk[:,0].min():.2f}/{bulk[:,0].max():.2f}  end min/max {bulk[:,5].min():.2f}/{bulk[:,5].max():.2f}")
print("  end of every other row block (us), in row order [z, W rows..., identity rows...]:", " ".join(f"{x:.0f}" for x in bulk[:, 5]))
print("  their starts of the last step (stamp 2: vector ops begin):", " ".join(f"{x:.0f}" for x in bulk[:, 2]))
nd = int((st[1 + RT:256, 0] > 0).sum())
if nd:
    d = us(st[1 + RT:1 + RT + nd])
    print(f"down-date WGs {nd}: start min/max {d[:,0].min():.2f}/{d[:,0].max():.2f}  end min/median/max {d[:,5].min():.2f}/{np.median(d[:,5]):.2f}/{d[:,5].max():.2f}")
    print(f"  last block done min/median/max {d[:,1].min():.2f}/{np.median(d[:,1]):.2f}/{d[:,1].max():.2f}; epilogue stores issued "
          f"{np.median(d[:,2] - d[:,1]):.2f} us later (median), vector ops + exit {np.median(d[:,5] - d[:,2]):.2f} us")
print("per-step trace of one row block (us): step-start, newest tile known, newest operand loaded, S_j done(barrier), bulk done, diag known, after X_j/accD, step end"
      "   [feed step: start, seen, loaded, stores issued, drained, flag set]")
for j in range(CB):
    a = us(st[256 + j])
    if st[256 + j, 0] <= 0:
        continue
    print(f"j={j:2d} " + " ".join(f"{a[i]:8.2f}" if st[256 + j, i] > 0 else "       -" for i in (0, 1, 7, 2, 4, 5, 3, 6)))

print("publishing wave of the same block (us): X_j complete (after barrier), stores issued, bulk done, stores drained + flag")
for j in range(CB):
    if st[320 + j, 0] <= 0:
        continue
    a = us(st[320 + j])
    print(f"j={j:2d} " + " ".join(f"{a[i]:8.2f}" for i in range(4)) + f"   | drain {a[3]-a[2]:5.2f}")

print("one down-date workgroup (pair 60), per column block (us): entered, [waited until], MFMAs issued | avail, fragments prefetched")
for j in range(CB):
    a = st[448 + j]
    if a[0] <= 0:
        continue
    print(f"j={j:2d} {us(a[0]):8.2f} " + (f"{us(a[1]):8.2f}" if a[1] > 0 else "       -") + f" {us(a[2]):8.2f} | {int(a[3])} {int(a[4])}")

