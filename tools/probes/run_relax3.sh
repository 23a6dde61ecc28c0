set -e
for n in 10000 20000 100000 200000; do
for cfg in "CS3_DUMMY=1" "CS3_RELAX_Z=0.7" "CS3_RELAX_Z=0.7 CS3_RELAX_W=6" "CS3_RELAX_Z=1.0"; do
echo "n=$n $cfg: $(env $cfg python tools/fused_timing.py $n 1 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["fused2"],4), round(d["split2"],4))')"
done; done
