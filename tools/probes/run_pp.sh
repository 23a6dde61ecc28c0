set -e
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t5.log 2>&1; tail -2 gpurun_out/t5.log
for i in 1 2; do
python bench.py --configs=5 --no-cpu-baseline --steps 200 > gpurun_out/pa$i.log 2>&1
CS3_NO_PANEL_PREFETCH=1 python bench.py --configs=5 --no-cpu-baseline --steps 200 > gpurun_out/pb$i.log 2>&1
done
python - <<PY
import json
for f in "ab":
    ds=[json.loads(open("gpurun_out/p%s%d.log"%(f,i)).read().strip().splitlines()[-1]) for i in (1,2)]
    print(f, [round(d["ms_per_step"],4) for d in ds], [round(d["phases"]["solve_ms"],4) for d in ds], [round(d["configs"]["5"]["factor_solve_ms"],3) for d in ds])
PY
