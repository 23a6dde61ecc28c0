set -e
for i in 1 2 3; do
python bench.py --configs=5 --no-cpu-baseline --steps 300 > gpurun_out/pa$i.log 2>&1
done
python - <<PY
import json
ds=[json.loads(open("gpurun_out/pa%d.log"%i).read().strip().splitlines()[-1]) for i in (1,2,3)]
print([round(d["ms_per_step"],4) for d in ds], [round(d["phases"]["factor_ms"],4) for d in ds], [round(d["configs"]["5"]["factor_solve_ms"],3) for d in ds])
PY
python tools/front_stamps.py 2>&1 | tail -1 | cut -c1-260
