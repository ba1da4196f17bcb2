#!/bin/bash
# usage (under gpurun): tools_prof_r03.sh <tag> [mesh_n] -- kernel-trace stats + separate PMC passes (HBM bytes; MFMA / busy)
# of the bench loop into gpurun_out/<tag>/.  The program comes directly behind `--` (no env/bash hop).
TAG=$1; MESH=${2:-512}; STEPS=${3:-50}
R=$GRAFT_REPO_ROOT
ARGS="$R/bench.py --steps $STEPS --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --mesh-n $MESH"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $ARGS > $R/gpurun_out/$TAG/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $ARGS > $R/gpurun_out/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $ARGS > $R/gpurun_out/$TAG/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/$TAG/pmc_mfma -- python3 $ARGS > $R/gpurun_out/$TAG/mfma.log 2>&1
rocprofv3 -L > $R/gpurun_out/$TAG/counters_list.txt 2>&1
ls $R/gpurun_out/$TAG/*/*/ 2>/dev/null | head -30
