set -e
run() { tag=$1; shift; env "$@" python bench.py --configs=5 --no-cpu-baseline --steps 20 > gpurun_out/rx_$tag.log 2>&1; }
run base CS3_DUMMY=1
run z05 CS3_RELAX_IL_Z=0.5
run z1 CS3_RELAX_IL_Z=1.0
run z2 CS3_RELAX_IL_Z=2.0
run z4 CS3_RELAX_IL_Z=4.0
run z1r12 CS3_RELAX_IL_Z=1.0 CS3_RELAX_IL_R=12
python - <<PY
import json
for f in ("base","z05","z1","z2","z4","z1r12"):
    d=json.loads(open("gpurun_out/rx_%s.log"%f).read().strip().splitlines()[-1])["configs"]["5"]
    print(f, round(d["factor_solve_ms"],3), "levels", d["levels"], "res", d["rel_residual"])
PY
