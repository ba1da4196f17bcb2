"""Summarise a tools_pmc.sh output directory into profiles/<tag>_summary.md (kernel stats + HBM bytes)."""
import csv, glob, sys, collections
tag, steps = sys.argv[1], int(sys.argv[2])
base = f"gpurun_out/{tag}"
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
stats = list(csv.DictReader(open(glob.glob(f"{base}/trace/*/*kernel_stats.csv")[0])))
nsteps = 2 + 2 * steps              # warm-up + rehearsal (prepare_steps) + timed
def counters(kind, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(glob.glob(f"{base}/pmc_{kind}/*/*counter_collection.csv")[0])):
        if r["Counter_Name"] == name:
            a = acc[short(r["Kernel_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
out = [f"# {tag}: rocprofv3 summary of `bench.py --steps {steps} --warmup 2` (N=512, nu=2, 1x MI355X)", "",
       "Kernel durations: `--kernel-trace --stats`. HBM bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes;",
       "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM).",
       "Durations include the dispatch gap to the previous kernel (back-to-back timestamps).", "",
       "| kernel | calls | avg us | per-step us | share | HBM read MB/launch (2xFETCH) | HBM write MB/launch |", "|---|---|---|---|---|---|---|"]
tot = 0.0
hbm_step = 0.0
for r in stats:
    n = short(r["Name"]); calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    per_step = float(r["TotalDurationNs"]) / 1e3 / nsteps; tot += per_step
    f = fetch.get(n); w = write.get(n)
    fr = f"{2 * f[0] / f[1] * 1024 / 1e6:.2f}" if f and f[1] else "-"
    wr = f"{w[0] / w[1] * 1024 / 1e6:.2f}" if w and w[1] else "-"
    if n.startswith("k_"):
        hbm_step += ((2 * f[0] / f[1] if f and f[1] else 0.0) + (w[0] / w[1] if w and w[1] else 0.0)) * 1024 * calls / nsteps
    out.append(f"| {n} | {calls} | {avg:.2f} | {per_step:.1f} | {r['Percentage']}% | {fr} | {wr} |")
out += ["", f"Sum of kernel time per step: {tot:.1f} us  ({nsteps} steps executed: warm-up, rehearsal, timed).",
        f"HBM/fabric bytes per step (all k_* kernels, 2xFETCH_SIZE + WRITE_SIZE): {hbm_step / 1e6:.1f} MB "
        f"(algorithmic B_alg = 3 D^2 w = 56.6 MB; the working set of ~90 MB sits in the 256 MB Infinity Cache)."]
import json
json.dump({"hbm_bytes_per_step": hbm_step, "source": f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
          open("profiles/traffic.json", "w"))
open(f"profiles/{tag}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
