#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# 1. kernel trace + stats of the bench command, 2. FETCH_SIZE pass, 3. WRITE_SIZE pass (PMC passes carry
# no trace domain other than --kernel-trace, as the pool requires).  Outputs land in gpurun_out/<tag>/.
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
tail -1 $OUT/trace.log | cut -c1-400
