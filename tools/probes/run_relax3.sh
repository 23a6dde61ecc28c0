set -e
run() { tag=$1; shift; env "$@" python bench.py --configs= --no-cpu-baseline --steps 200 > gpurun_out/r3_$tag.log 2>&1; }
run base CS3_DUMMY=1
run nbk8 CS3_NBK=8
run nbk32 CS3_NBK=32
run norootpipe CS3_NO_ROOT_PIPE=1
run k1 CS3_ROOT_K=1
run k3 CS3_ROOT_K=3
run absorb CS3_ABSORB=1
run fl9 CS3_FORK_LEVEL=9
run fl8 CS3_FORK_LEVEL=8
run base2 CS3_DUMMY=1
python - <<PY
import json
for f in ("base","nbk8","nbk32","norootpipe","k1","k3","absorb","fl9","fl8","base2"):
    d=json.loads(open("gpurun_out/r3_%s.log"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), "factor", round(d["phases"]["factor_ms"],3), "solve", round(d["phases"]["solve_ms"],3))
PY
