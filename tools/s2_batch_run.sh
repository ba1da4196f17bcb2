#!/bin/bash
# (under gpurun) eight problems in flight on one GPU: fused sweep (default) against sweep alone + k_downdate_big (PNMOL_HIP_DD_BIG=1)
OUT=gpurun_out/$1; mkdir -p $OUT
for v in "PNMOL_X=0" "PNMOL_HIP_DD_BIG=1" "PNMOL_HIP_DD_BIG=1 PNMOL_HIP_SWEEP_XL=0"; do
  echo "== $v" >> $OUT/batch.log
  env $v timeout -k 10 300 python tools/batch_modes.py 8 100 >> $OUT/batch.log 2>&1 || exit 1
done
cat $OUT/batch.log
