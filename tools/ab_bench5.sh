BASE=$(realpath csparse3_amd/libcs3_base.so)
for i in 1 2 3; do
  for which in base tree; do
    if [ $which = base ]; then export CS3_LIB_PATH=$BASE; else unset CS3_LIB_PATH; fi
    c5=$(python3 tools/bench_configs.py --nmat 512 --rhs 128 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f" % (d["config5_slice"]["ms"]))')
    echo "$which config5(512) ms $c5"
  done
done
