set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --configs= --no-cpu-baseline > gpurun_out/d0.log 2>&1
CS3_NO_ABSORB=1 python bench.py --configs= --no-cpu-baseline > gpurun_out/d1.log 2>&1
CS3_ROOT_K=2 python bench.py --configs= --no-cpu-baseline > gpurun_out/d2.log 2>&1
CS3_ROOT_K=2 CS3_NO_ABSORB=1 python bench.py --configs= --no-cpu-baseline > gpurun_out/d3.log 2>&1
CS3_ROOT_K=1 python bench.py --configs= --no-cpu-baseline > gpurun_out/d4.log 2>&1
CS3_ROOT_K=0 python bench.py --configs= --no-cpu-baseline > gpurun_out/d5.log 2>&1
python - <<PY
import json
for f in ("d0","d1","d2","d3","d4","d5"):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["phases"]["factor_ms"], d["phases"]["solve_ms"], d["phases"]["rel_residual"])
PY
