#!/usr/bin/env python3
"""Per-kernel shares of SQ_WAVE_CYCLES from the counter pass of tools/pmc_cfg5.sh:
    python tools/sq_summary.py gpurun_out/r02_cfg5_sq > profiles/r02_cfg5_sq_counters.txt"""
import collections, csv, glob, os, sys

src = sys.argv[1]
f = max(glob.glob(os.path.join(src, "sq", "*", "*_counter_collection.csv")), key=os.path.getmtime)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs3::", "")
    tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], n)
    if key not in seen:
        seen.add(key); calls[n] += 1
print("# SQ counters per kernel, config-5 slice (512 SPD 5000x5000 matrices, Cholesky factor + solve), tools/pmc_cfg5.sh")
print("# rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU")
print("# shares are of SQ_WAVE_CYCLES: parked = SQ_WAIT_ANY (s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stall), active = SQ_ACTIVE_INST_ANY")
print("%-30s %6s %14s %7s %7s %7s %9s %12s" % ("kernel", "calls", "wave_cycles(M)", "parked", "stall", "active", "valu_act", "valu/wave"))
for n in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0.0))[:20]:
    c = tot[n]
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print("%-30s %6d %14.1f %6.0f%% %6.0f%% %6.0f%% %8.0f%% %12.0f" % (
        n[:30], calls[n], wc / 1e6, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 0), 1.0)))
