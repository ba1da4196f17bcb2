#!/bin/bash
# (under gpurun) the square-root suites, then the sqrt step timed (30 steps) in fp64 / fp32 under the QR switches of the library:
# PNMOL_QR_OWNER (a column's dlarfg formed once, by its owners), PNMOL_QR_PRE (a panel's last trailing update in the launch of the next
# panel's first factorisation), PNMOL_QR_INLOOP (V^T V and T formed inside the column loop), PNMOL_QR_FUSE (apply of a level + factor
# of the next in one launch)
OUT=gpurun_out/$1; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_sqrt_fp32.py tests/test_gpu_sqrtform.py tests/test_gpu_sqrt.py tests/test_gpu_abi_lifetime.py -q > $OUT/sqrt_tests.log 2>&1
rc=$?; tail -25 $OUT/sqrt_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out"; exit 1; fi
for dt in f64 f32; do
  for v in "PNMOL_X=0" "PNMOL_QR_OWNER=0" "PNMOL_QR_PRE=1" "PNMOL_QR_INLOOP=0" "PNMOL_QR_INLOOP=0 PNMOL_QR_FUSE=0"; do
    echo "== dtype $dt $v" >> $OUT/bench_sqrt.log
    env $v timeout -k 10 300 python tools/bench_sqrt.py 512 30 $dt >> $OUT/bench_sqrt.log 2>&1 || exit 1
  done
done
cut -c1-120 $OUT/bench_sqrt.log
