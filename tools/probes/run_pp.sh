set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_factor_fixtures.py -x -q -m gpu -k "fused_root or fixture or denseblock or config3 or fresh or batched" > gpurun_out/t5.log 2>&1; tail -2 gpurun_out/t5.log
for i in 1 2 3; do
python bench.py --configs=5 --no-cpu-baseline --steps 300 > gpurun_out/pa$i.log 2>&1
done
python - <<PY
import json
ds=[json.loads(open("gpurun_out/pa%d.log"%i).read().strip().splitlines()[-1]) for i in (1,2,3)]
print([round(d["ms_per_step"],4) for d in ds], [round(d["phases"]["factor_ms"],4) for d in ds], [round(d["configs"]["5"]["factor_solve_ms"],3) for d in ds])
PY
python tools/front_stamps.py 2>&1 | tail -1
