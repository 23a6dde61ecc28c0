#!/bin/bash
# rocprofv3 kernel trace of the config-4 solve (factor once, many RHS): tools/profile_cfg4.sh <tag> <rhs>
set -e
TAG=${1:-cfg4}; RHS=${2:-128}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_cfg4.py --rhs $RHS --reps 10 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-600
