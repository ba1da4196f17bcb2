#!/bin/bash
# (under gpurun) bench_sqrt.py 512 30 in fp64 / fp32 under a list of environment variants: bash tools/s2_sqrt_ab.sh OUT "VAR=1" "VAR=0 OTHER=1" ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for dt in f64 f32; do
  for v in "$@"; do
    echo "== dtype $dt $v" >> $OUT/bench_sqrt.log
    env $v timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
  done
done
cut -c1-120 $OUT/bench_sqrt.log
