#!/bin/bash
# (under gpurun) the square-root suites on the built library, then bench_sqrt.py 512 30 (fp64, fp32) alternating between a saved
# library (lib/libpnmol_hip_prev.so) and the built one, twice: same-box A/B of a change to the QR kernels
OUT=gpurun_out/$1; mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/pnmol-experiments_amd/lib/libpnmol_hip_prev.so
timeout -k 10 900 python -m pytest tests/test_gpu_sqrt_fp32.py tests/test_gpu_sqrtform.py tests/test_gpu_sqrt.py -q > $OUT/sqrt_tests.log 2>&1
rc=$?; tail -4 $OUT/sqrt_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out"; exit 1; fi
for rep in 1 2; do
  for dt in f64 f32; do
    echo "== $dt prev" >> $OUT/bench_sqrt.log
    PNMOL_HIP_LIB=$PREV timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
    echo "== $dt cur" >> $OUT/bench_sqrt.log
    timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
  done
done
cut -c1-110 $OUT/bench_sqrt.log
