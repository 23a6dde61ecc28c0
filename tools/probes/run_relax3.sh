set -e
run() { tag=$1; shift; env "$@" python bench.py --configs= --no-cpu-baseline --steps 100 > gpurun_out/r3_$tag.log 2>&1; }
run off CS3_NO_HEIGHT_PASS=1
run zc1 CS3_RELAX_ZC=1.0
run zc2 CS3_RELAX_ZC=2.0
run zc4 CS3_RELAX_ZC=4.0
run zc8 CS3_RELAX_ZC=8.0
python - <<PY
import json
for f in ("off","zc1","zc2","zc4","zc8"):
    d=json.loads(open("gpurun_out/r3_%s.log"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), "levels", d["config"]["levels"], "supernodes", d["config"]["supernodes"], "factor", round(d["phases"]["factor_ms"],3), "solve", round(d["phases"]["solve_ms"],3))
PY
for n in 10000 20000 100000 200000; do
for cfg in "CS3_NO_HEIGHT_PASS=1" "CS3_RELAX_ZC=2.0" "CS3_RELAX_ZC=4.0"; do
echo "n=$n $cfg: $(env $cfg python tools/fused_timing.py $n 1 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["fused2"],4), round(d["split2"],4))')"
done; done
