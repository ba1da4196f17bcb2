#!/bin/bash
# usage (under gpurun): tools_prof_r03_cmd.sh <tag> <script.py> [args...] -- kernel-trace stats + PMC passes of any tool
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
echo "python3 $*" > $R/gpurun_out/$TAG/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/$@ > $R/gpurun_out/$TAG/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/$@ > $R/gpurun_out/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/$@ > $R/gpurun_out/$TAG/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/$TAG/pmc_mfma -- python3 $R/$@ > $R/gpurun_out/$TAG/mfma.log 2>&1
