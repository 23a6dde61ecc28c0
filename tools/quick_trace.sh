#!/bin/bash
# Kernel trace + stats of a short bench run on the GPU box:  tools/quick_trace.sh <tag> [extra bench args]
# Prints the per-kernel stats (name, calls, average ns) of the hot path.
set -e
TAG=${1:-qt}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs= "$@" > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print("%-90s calls %6s avg %10.0f ns total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
tail -1 $OUT/trace.log | cut -c1-300
