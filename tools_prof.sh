#!/bin/bash
# usage: tools_prof.sh <tag> [steps]   -- run under gpurun: kernel-trace stats of bench.py into gpurun_out/<tag>/
TAG=$1; STEPS=${2:-50}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/$TAG/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel total per step: %.1f us"%(tot/1e3/($STEPS+2)))
for r in rows[:14]: print("%-50s calls=%6s avg_us=%8.2f per_step_us=%8.1f pct=%s"%(r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/($STEPS+2), r['Percentage']))
PY
