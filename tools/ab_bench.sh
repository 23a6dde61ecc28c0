#!/bin/bash
# A/B on ONE box: the library in the tree against another build of it (CS3_LIB_PATH), alternating, config 3 and the config-5 slice.
#   tools/ab_bench.sh csparse3_amd/libcs3_base.so [rounds]
BASE=$(realpath ${1:-csparse3_amd/libcs3_base.so}); N=${2:-3}
for i in $(seq $N); do
  for which in base tree; do
    if [ $which = base ]; then export CS3_LIB_PATH=$BASE; else unset CS3_LIB_PATH; fi
    c3=$(python3 bench.py --no-cpu-baseline --configs= 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.4f" % json.loads(sys.stdin.read())["ms_per_step"])')
    c5=$(python3 tools/bench_configs.py --nmat 512 --rhs 128 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f %.4f" % (d["config5_slice"]["ms"], d["config4_slice"]["ms"]))')
    echo "$which config3 ms_per_step $c3  config5(512) / config4(128 rhs) ms $c5"
  done
done
