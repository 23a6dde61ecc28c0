set -e
for i in 1 2 3; do
python bench.py --configs=5 --no-cpu-baseline --steps 50 > gpurun_out/a$i.log 2>&1
CS3_IL_SWEEP_RMAX=0 python bench.py --configs=5 --no-cpu-baseline --steps 50 > gpurun_out/b$i.log 2>&1
CS3_IL_SWEEP_RMAX=24 python bench.py --configs=5 --no-cpu-baseline --steps 50 > gpurun_out/c$i.log 2>&1
done
python - <<PY
import json
for f in ("a","b","c"):
    print(f, [round(json.loads(open("gpurun_out/%s%d.log"%(f,i)).read().strip().splitlines()[-1])["configs"]["5"]["factor_solve_ms"],3) for i in (1,2,3)])
PY
