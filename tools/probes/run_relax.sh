set -e
run() { tag=$1; shift; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 > gpurun_out/rx_$tag.log 2>&1; }
run base CS3_DUMMY=1
run m32 CS3_MIX_RMAX=32
run m48 CS3_MIX_RMAX=48
run base2 CS3_DUMMY=1
run m32b CS3_MIX_RMAX=32
python - <<PY
import json
for f in ("base","m32","m48","base2","m32b"):
    d=json.loads(open("gpurun_out/rx_%s.log"%f).read().strip().splitlines()[-1])["configs"]["5"]
    print(f, round(d["factor_solve_ms"],3), "res", d["rel_residual"])
PY
