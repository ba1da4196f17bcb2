#!/bin/bash
# (under gpurun) tree fan-in 8 (the built library) against 4 (lib/libpnmol_hip_fan4.so: hipcc ... -DPNMOL_QR_FAN=4): the square-root
# suites under the fan-4 library, then bench_sqrt.py 512 30 in fp64 / fp32 with both
OUT=gpurun_out/$1; mkdir -p $OUT
F4=$GRAFT_REPO_ROOT/pnmol-experiments_amd/lib/libpnmol_hip_fan4.so
PNMOL_HIP_LIB=$F4 timeout -k 10 900 python -m pytest tests/test_gpu_sqrt_fp32.py tests/test_gpu_sqrtform.py tests/test_gpu_sqrt.py -q > $OUT/sqrt_tests_fan4.log 2>&1
rc=$?; tail -12 $OUT/sqrt_tests_fan4.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out"; exit 1; fi
for dt in f64 f32; do
  echo "== dtype $dt fan 8" >> $OUT/bench_sqrt.log
  timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
  echo "== dtype $dt fan 4" >> $OUT/bench_sqrt.log
  PNMOL_HIP_LIB=$F4 timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
done
cut -c1-120 $OUT/bench_sqrt.log
