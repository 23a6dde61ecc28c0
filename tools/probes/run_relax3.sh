set -e
run() { tag=$1; shift; env "$@" python bench.py --configs= --no-cpu-baseline --steps 100 > gpurun_out/r3_$tag.log 2>&1; }
run t256 CS3_DUMMY=1
run t128 CS3_MIX_THREADS=128
run t64 CS3_MIX_THREADS=64
run t256b CS3_DUMMY=1
run t128b CS3_MIX_THREADS=128
python - <<PY
import json
for f in ("t256","t128","t64","t256b","t128b"):
    d=json.loads(open("gpurun_out/r3_%s.log"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), "factor", round(d["phases"]["factor_ms"],3), "solve", round(d["phases"]["solve_ms"],3))
PY
