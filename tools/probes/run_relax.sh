set -e
run() { tag=$1; shift; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 > gpurun_out/rx_$tag.log 2>&1; }
run fl6 CS3_FORK_LEVEL=6
run fl7 CS3_FORK_LEVEL=7
run fl8 CS3_FORK_LEVEL=8
run fl9 CS3_FORK_LEVEL=9
run fl7n8 CS3_FORK_LEVEL=7 CS3_NBK=8
run fl8n8 CS3_FORK_LEVEL=8 CS3_NBK=8
run base CS3_DUMMY=1
python - <<PY
import json
for f in ("fl6","fl7","fl8","fl9","fl7n8","fl8n8","base"):
    d=json.loads(open("gpurun_out/rx_%s.log"%f).read().strip().splitlines()[-1])["configs"]["5"]
    print(f, round(d["factor_solve_ms"],3), "levels", d["levels"])
PY
