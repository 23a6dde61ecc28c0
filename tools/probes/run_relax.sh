set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_big or own_size or interleaved" > gpurun_out/t5.log 2>&1; tail -2 gpurun_out/t5.log
run() { tag=$1; mats=$2; shift 2; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 --c5-mats $mats > gpurun_out/rx_$tag.log 2>&1; }
for mats in 64 256 512; do
run a_$mats $mats CS3_DUMMY=1
run b_$mats $mats CS3_DUMMY=1
done
python - <<PY
import json
for mats in (64,256,512):
    print(mats, [round(json.loads(open("gpurun_out/rx_%s_%d.log"%(f,mats)).read().strip().splitlines()[-1])["configs"]["5"]["factor_solve_ms"],3) for f in "ab"])
PY
