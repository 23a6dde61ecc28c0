set -e
run() { tag=$1; mats=$2; shift 2; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 --c5-mats $mats > gpurun_out/rx_$tag.log 2>&1; }
for mats in 32 64 128 256; do
run a_$mats $mats CS3_DUMMY=1
run b_$mats $mats CS3_WG_MIN_BATCH=100000
done
python - <<PY
import json
for mats in (32,64,128,256):
    print(mats, [round(json.loads(open("gpurun_out/rx_%s_%d.log"%(f,mats)).read().strip().splitlines()[-1])["configs"]["5"]["factor_solve_ms"],3) for f in "ab"])
PY
