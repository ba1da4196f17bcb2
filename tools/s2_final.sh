#!/bin/bash
# (under gpurun) the round's closing run at the final code: full GPU suite, bench line, rocprofv3 passes of the bench workload and of
# the square-root step in fp64 / fp32
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log
timeout -k 10 400 python bench.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
grep "^{" $OUT/bench.log > $OUT/bench_line.json; cut -c1-400 $OUT/bench_line.json
bash tools/prof_r03.sh $1/n512 512 50 > $OUT/prof_n512.log 2>&1 || exit 1
bash tools/prof_r03_cmd.sh $1/sqrt_f64 tools/bench_sqrt.py 512 30 f64 > $OUT/prof_sqrt64.log 2>&1 || exit 1
bash tools/prof_r03_cmd.sh $1/sqrt_f32 tools/bench_sqrt.py 512 30 f32 > $OUT/prof_sqrt32.log 2>&1 || exit 1
ls $OUT $OUT/n512 $OUT/sqrt_f32
