set -e
for i in 1 2 3; do
python bench.py --configs= --no-cpu-baseline --steps 400 > gpurun_out/a$i.log 2>&1
CS3_NO_ABSORB=1 python bench.py --configs= --no-cpu-baseline --steps 400 > gpurun_out/b$i.log 2>&1
CS3_ROOT_K=2 CS3_NO_ABSORB=1 python bench.py --configs= --no-cpu-baseline --steps 400 > gpurun_out/c$i.log 2>&1
done
python - <<PY
import json
for f in ("a","b","c"):
    print(f, [round(json.loads(open("gpurun_out/%s%d.log"%(f,i)).read().strip().splitlines()[-1])["ms_per_step"],4) for i in (1,2,3)])
PY
