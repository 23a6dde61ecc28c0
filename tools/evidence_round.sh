#!/bin/bash
# All GPU-side evidence of a round in one gpurun call:  tools/evidence_round.sh r02
# (config 3 trace + PMC passes, config 4 / 5 kernel stats, SQ counters of config 5, 2-rank gloo rehearsal, bench JSON)
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG
bash tools/profile_cfg4.sh ${TAG}_cfg4_1024 1024
bash tools/profile_cfg4.sh ${TAG}_cfg4_128 128
bash tools/profile_cfg5.sh ${TAG}_cfg5_512 512
bash tools/pmc_cfg5.sh ${TAG}_cfg5_sq 512
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 \
    --backend gloo --one-device --no-cpu-baseline > gpurun_out/${TAG}_bench_gloo2.log 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
