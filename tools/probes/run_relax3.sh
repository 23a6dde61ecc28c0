set -e
run() { tag=$1; shift; env "$@" python bench.py --configs= --no-cpu-baseline --steps 100 > gpurun_out/r3_$tag.log 2>&1; }
run base CS3_DUMMY=1
run wm12 CS3_RELAX_WMAX=12
run wm16 CS3_RELAX_WMAX=16
run wm20 CS3_RELAX_WMAX=20
python - <<PY
import json
for f in ("base","wm12","wm16","wm20"):
    d=json.loads(open("gpurun_out/r3_%s.log"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), "levels", d["config"]["levels"], "supernodes", d["config"]["supernodes"], "factor", round(d["phases"]["factor_ms"],3), "solve", round(d["phases"]["solve_ms"],3))
PY
