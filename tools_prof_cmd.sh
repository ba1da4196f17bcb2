#!/bin/bash
# usage: tools_prof_cmd.sh <tag> <python script> [args...]  -- run under gpurun: kernel-trace stats of a tool script
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/"$@" > $R/gpurun_out/$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/$TAG/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]: print("%-50s calls=%6s avg_us=%8.2f total_ms=%8.2f pct=%s"%(r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
