#!/bin/bash
# (under gpurun) the square-root suites, then the sqrt step timed in fp64 / fp32 with and without the fused apply+factor launch
OUT=gpurun_out/$1; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_sqrt_fp32.py tests/test_gpu_sqrtform.py tests/test_gpu_sqrt.py -q > $OUT/sqrt_tests.log 2>&1
rc=$?; tail -25 $OUT/sqrt_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out"; exit 1; fi
for dt in f64 f32; do
  for fuse in 1 0; do
    echo "== dtype $dt fuse $fuse" >> $OUT/bench_sqrt.log
    PNMOL_QR_FUSE=$fuse timeout -k 10 300 python tools/bench_sqrt.py 512 10 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
  done
done
cat $OUT/bench_sqrt.log
