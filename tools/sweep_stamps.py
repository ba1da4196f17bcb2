"""Timeline of one k_sweep launch (library built with -DPNMOL_SWEEP_STAMP, loaded through PNMOL_HIP_LIB)."""
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
out, info, err = flt.step(dev, bench.DT)     # one eager step: the stamps of its k_sweep
st = flt.debug_read(5, 512 * 8).reshape(512, 8)
mp = flt.dims()["mp"]
CB = mp // 32
t0 = st[:, 0][st[:, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
print("chain WG: start, acc-done(last step), diag-seen, factor-start, published | deltas: wait, trsm+syrk, factor")
prev = 0.0
for I in range(CB):
    a = us(st[I])
    print(f"I={I:2d} start {a[0]:7.2f} accdone {a[1]:7.2f} diagseen {a[2]:7.2f} fstart {a[3]:7.2f} pub {a[4]:7.2f} | "
          f"hop {a[2]-prev:5.2f} prep {a[3]-a[2]:5.2f} factor {a[4]-a[3]:5.2f}  panel {a[4]-prev:5.2f}")
    prev = a[4]
dp = flt.dims()["dp"]
RT = 2 * CB + 3 * dp // 32 + 1
ends = us(st[CB:RT, 5])
starts = us(st[CB:RT, 0])
print(f"bulk WGs {RT-CB}: start min/max {starts.min():.2f}/{starts.max():.2f}  end min/max {ends.min():.2f}/{ends.max():.2f}")
order = np.argsort(ends)[::-1][:6]
print("  latest bulk blocks (physical index: loop end / r^T seen / end):", ", ".join(f"{CB + int(i)}: {us(st[CB + int(i)])[1]:.1f}/{us(st[CB + int(i)])[2]:.1f}/{ends[i]:.1f}" for i in order))
print("  first W-row blocks:", ", ".join(f"{CB + 1 + i}: {us(st[CB + 1 + i])[1]:.1f}/{us(st[CB + 1 + i])[2]:.1f}/{ends[1 + i]:.1f}" for i in range(4)), f"; r^T block {CB}: end {ends[0]:.1f}")
nd = int((st[RT:256, 0] > 0).sum())
if nd:
    e2, s2 = us(st[RT:RT + nd, 5]), us(st[RT:RT + nd, 0])
    print(f"down-date WGs {nd}: start min/max {s2.min():.2f}/{s2.max():.2f}  end min/median/max {e2.min():.2f}/{np.median(e2):.2f}/{e2.max():.2f}")
    m1, m2 = us(st[RT:RT + nd, 1]), us(st[RT:RT + nd, 2])
    print(f"  last block done min/median/max {m1.min():.2f}/{np.median(m1):.2f}/{m1.max():.2f}; epilogue stores issued "
          f"{np.median(m2 - m1):.2f} us later (median), vector ops + exit {np.median(e2 - m2):.2f} us")

print("per-step trace of WG 16 (us): step-start, row[j] seen, S_j done(barrier), row[j+1] seen, partial done, diag seen, step end")
for j in range(CB - 1):
    a = us(st[256 + j])
    print(f"j={j:2d} " + " ".join(f"{x:8.2f}" for x in a[:7]) + "   | " + " ".join(f"{a[i+1]-a[i]:5.2f}" for i in range(6)))
