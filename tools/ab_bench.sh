#!/bin/bash
# Same-box A/B of the headline step: the library of the previous round (lib/libpnmol_hip_prev.so, not tracked) against the
# current build.  usage (on the GPU box): bash tools/ab_bench.sh OUTDIR [variants...]   (default: prev cur)
out=gpurun_out/$1; shift
mkdir -p $out
vs="${@:-prev cur}"
for v in $vs; do
  if [ $v = cur ]; then unset PNMOL_HIP_LIB; else export PNMOL_HIP_LIB=$GRAFT_REPO_ROOT/pnmol-experiments_amd/lib/libpnmol_hip_$v.so; fi
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-extras --repeats 5 > $out/bench_$v.log 2>&1
  python - $out/bench_$v.log $v <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l)
        print(sys.argv[2], j["value"], j["ms_per_step"], j.get("roofline", {}).get("achieved"))
        break
else:
    print(sys.argv[2], "no bench line:", open(sys.argv[1]).read()[-600:])
PY
done
