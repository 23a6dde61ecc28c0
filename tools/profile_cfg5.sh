#!/bin/bash
# rocprofv3 kernel stats of the config-5 slice (batch of SPD matrices, one pattern): tools/profile_cfg5.sh <tag> <nmat>
set -e
TAG=${1:-cfg5}; NMAT=${2:-256}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_configs.py --rhs 16 --nmat $NMAT --reps 10 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-600
