set -e
timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/t5.log 2>&1
run() { tag=$1; shift; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 > gpurun_out/rx_$tag.log 2>&1; }
run base CS3_DUMMY=1
run il12 CS3_IL_RMAX=12
run il20 CS3_IL_RMAX=20
run il24 CS3_IL_RMAX=24
run nb8 CS3_NBK=8
run nb32 CS3_NBK=32
run r24 CS3_RELAX_R=24
run z2 CS3_RELAX_Z2=0.1
tail -2 gpurun_out/t5.log
python - <<PY
import json
for f in ("base","il12","il20","il24","nb8","nb32","r24","z2"):
    d=json.loads(open("gpurun_out/rx_%s.log"%f).read().strip().splitlines()[-1])["configs"]["5"]
    print(f, round(d["factor_solve_ms"],3), "levels", d["levels"], "res", d["rel_residual"])
PY
