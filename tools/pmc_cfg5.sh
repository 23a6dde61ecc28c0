#!/bin/bash
# SQ counters of the config-5 slice per kernel (one pass, 8 SQ slots): tools/pmc_cfg5.sh <tag> <nmat>
set -e
TAG=${1:-pmc5}; NMAT=${2:-512}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 tools/bench_configs.py --rhs 16 --nmat $NMAT --reps 3 > $OUT/sq.log 2>&1
ls $OUT/sq/*/ | head
