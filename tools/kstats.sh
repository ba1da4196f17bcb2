#!/bin/bash
# rocprofv3 kernel statistics of a tool script, printed as "kernel calls avg_us" (on the GPU box): bash tools/kstats.sh OUT tools/x.py [args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/$@ > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"{n:50s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:12.1f} us")
PY
