#!/bin/bash
# 8 problems in flight on one GPU under the sweep-kernel switches of the library (bench.py --problems-per-gpu 8).
# usage (on the GPU box): bash tools/batch8_variants.sh OUTDIR
out=gpurun_out/$1; mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 3 --problems-per-gpu 8 > $out/$name.log 2>&1
  python - $out/$name.log $name <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l); print(sys.argv[2], round(j["value"]), "steps/s", j["ms_per_step"], j["config"].get("sweep_layout"))
        break
else:
    print(sys.argv[2], "no line", open(sys.argv[1]).read()[-400:])
PY
}
run default PNMOL_X=0
run rl0 PNMOL_HIP_SWEEP_RL=0
run xl0 PNMOL_HIP_SWEEP_XL=0
run sweep1 PNMOL_HIP_SWEEP=1
