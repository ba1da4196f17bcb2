#!/bin/bash
# usage (under gpurun): tools_pmc.sh <tag> [steps] -- kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
TAG=$1; STEPS=${2:-20}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG.trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG.write.log 2>&1
ls $R/gpurun_out/$TAG/*/*/ | head -20
