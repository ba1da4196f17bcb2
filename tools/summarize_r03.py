"""Summarise a tools_prof_r03.sh output directory (gpurun_out/<tag>/) into profiles/<name>_summary.md:
kernel durations (--kernel-trace --stats), HBM bytes (separate --pmc FETCH_SIZE / WRITE_SIZE passes) and the MFMA
counters (one more --pmc pass).  usage: tools/summarize_r03.py <tag> <name> <steps> [--traffic-json] [--nsteps N]
(N: filter steps the profiled command executed in all, when it is not bench.py's 2 + 2 * steps: the VERDICT of round 2 found a
summary that divided by the wrong count)"""
import csv, glob, sys, collections, json
tag, name, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
base = f"gpurun_out/{tag}"
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
stats = list(csv.DictReader(open(glob.glob(f"{base}/trace/*/*kernel_stats.csv")[0])))
nsteps = 2 + 2 * steps              # bench.py: warm-up + rehearsal (prepare_steps) + timed
if "--nsteps" in sys.argv:
    nsteps = int(sys.argv[sys.argv.index("--nsteps") + 1])
def counters(kind):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(glob.glob(f"{base}/pmc_{kind}/*/*counter_collection.csv")[0])):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
fetch, write, mfma = counters("fetch"), counters("write"), counters("mfma")
avg = lambda d, k, c: (d[k][c][0] / d[k][c][1]) if k in d and c in d[k] and d[k][c][1] else None
out = [f"# {name}: rocprofv3 summary of `{open(base + '/cmd.txt').read().strip() if glob.glob(base + '/cmd.txt') else 'bench.py'}` (1x MI355X)", "",
       "Kernel durations: `--kernel-trace --stats`.  HBM bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; both are in KiB and",
       "FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md, HBM).  MFMA: one more pass with",
       "`--pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA`: executed flops =",
       "MOPS_F64 x 512; MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs) (nominal clock: a lower bound).", "",
       "| kernel | calls | avg us | per-step us | share | HBM read MB (2xFETCH) | HBM write MB | MFMA instr | executed GFLOP (fp64 MFMA) | MFMA pipe busy |", "|---|---|---|---|---|---|---|---|---|---|"]
tot = hbm_step = 0.0
for r in stats:
    n = short(r["Name"]); calls = int(r["Calls"]); a_us = float(r["AverageNs"]) / 1e3
    per_step = float(r["TotalDurationNs"]) / 1e3 / nsteps; tot += per_step
    f, w = avg(fetch, n, "FETCH_SIZE"), avg(write, n, "WRITE_SIZE")
    mi, mo, mb = avg(mfma, n, "SQ_INSTS_MFMA"), avg(mfma, n, "SQ_INSTS_VALU_MFMA_MOPS_F64"), avg(mfma, n, "SQ_VALU_MFMA_BUSY_CYCLES")
    if n.startswith("k_"):
        hbm_step += ((2 * f if f else 0.0) + (w if w else 0.0)) * 1024 * calls / nsteps
    fmt = lambda x, s=1.0, p=2: f"{x * s:.{p}f}" if x is not None else "-"
    util = f"{100 * mb / (a_us * 1e-6 * 2.4e9 * 1024):.1f} %" if mb else "-"
    out.append(f"| {n} | {calls} | {a_us:.2f} | {per_step:.1f} | {r['Percentage']}% | {fmt(f, 2 * 1024 / 1e6)} | {fmt(w, 1024 / 1e6)} | "
               f"{fmt(mi, 1, 0)} | {fmt(mo, 512 / 1e9, 3)} | {util} |")
out += ["", f"Sum of kernel time per step: {tot:.1f} us  ({nsteps} steps executed: warm-up, rehearsal, timed).",
        f"HBM/fabric bytes per step (all k_* kernels, 2xFETCH_SIZE + WRITE_SIZE): {hbm_step / 1e6:.1f} MB."]
if "--traffic-json" in sys.argv:
    json.dump({"hbm_bytes_per_step": hbm_step, "source": f"profiles/{name}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
              open("profiles/traffic.json", "w"))
open(f"profiles/{name}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
