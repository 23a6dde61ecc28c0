set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_factor_fixtures.py -x -q -m gpu -k "interleaved or batch or own_size or config5" > gpurun_out/t5.log 2>&1; tail -2 gpurun_out/t5.log
for i in 1 2 3; do python bench.py --configs=5 --no-cpu-baseline --steps 20 > gpurun_out/rx_a$i.log 2>&1; done
python - <<PY
import json
print([round(json.loads(open("gpurun_out/rx_a%d.log"%i).read().strip().splitlines()[-1])["configs"]["5"]["factor_solve_ms"],3) for i in (1,2,3)])
PY
