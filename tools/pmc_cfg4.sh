#!/bin/bash
# HBM traffic of the config-4 solve per kernel (FETCH_SIZE and WRITE_SIZE in passes of their own, --kernel-trace only):
#   tools/pmc_cfg4.sh <tag> <rhs>   then   python tools/pmc_cfg4_summary.py gpurun_out/<tag> <reps>
set -e
TAG=${1:-pmc4}; RHS=${2:-1024}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/bench_cfg4.py --rhs $RHS --reps 5 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/bench_cfg4.py --rhs $RHS --reps 5 > $OUT/write.log 2>&1
tail -1 $OUT/write.log | cut -c1-300
