set -e
for fl in 2 3 4 5 6; do
CS3_FORK_LEVEL=$fl python bench.py --configs=5 --no-cpu-baseline > gpurun_out/f$fl.log 2>&1
done
python bench.py --configs=5 --no-cpu-baseline > gpurun_out/fd.log 2>&1
python - <<PY
import json
for f in ("f2","f3","f4","f5","f6","fd"):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["configs"]["5"]["factor_solve_ms"])
PY
