set -e
run() { tag=$1; rhs=$2; shift 2; env "$@" python tools/bench_cfg4.py --rhs $rhs --reps 20 > gpurun_out/c4_$tag.log 2>&1; }
for i in 1 2; do
for rhs in 1024 128 256; do
run base_${rhs}_$i $rhs CS3_DUMMY=1
run nosplit_${rhs}_$i $rhs CS3_NO_SPLIT16=1
run nofork_${rhs}_$i $rhs CS3_SOLVE_FORK=0
run both_${rhs}_$i $rhs CS3_SOLVE_FORK=0 CS3_NO_SPLIT16=1
done; done
python - <<PY
import json
for rhs in (1024,256,128):
    for f in ("base","nosplit","nofork","both"):
        print(rhs,f,[round(json.loads(open("gpurun_out/c4_%s_%d_%d.log"%(f,rhs,i)).read().strip().splitlines()[-1])["ms"],4) for i in (1,2)])
PY
