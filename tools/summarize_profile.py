#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into the committed evidence under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --stats as is), <tag>_pmc_summary.json (per-kernel FETCH_SIZE /
WRITE_SIZE per step) and <tag>_traffic.json (what bench.py reports as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 86          # 1 capture + 5 warmup + 40 timed + 40 phase-pass steps
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):          # gpurun merges every call's outputs into the same directory: take the latest run's file
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("cs3::", "")


def per_kernel(kind):
    f = newest(os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv"))
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        tot[n] += float(r["Counter_Value"]); cnt[n] += 1
    return {n: {"calls_per_step": cnt[n] / steps, "kb_per_step": tot[n] / steps} for n in tot}


fetch, write = per_kernel("fetch"), per_kernel("write")
factor_kernels = [n for n in fetch if n.startswith("k_front") or n.startswith("k_big") or n.startswith("k_sub_factor")]
solve_kernels = [n for n in fetch if n.startswith("k_fwd") or n.startswith("k_bwd") or n.startswith("k_permute") or
                 n.startswith("k_sub_fwd") or n.startswith("k_sub_bwd")]
kb = lambda table, names: sum(table.get(n, {"kb_per_step": 0.0})["kb_per_step"] for n in names)
# FETCH_SIZE / WRITE_SIZE are KB (x 1024).  Calibration on known byte counts in this very profile:
#   __amd_rocclr_copyBuffer (Ax + b, 16 B/lane streaming): FETCH_SIZE reads exactly 1/2 of the bytes copied,
#   WRITE_SIZE reads them exactly -- as MI355X_MICROARCH.md "HBM" says.  The guide's correction (double the
#   fetch side) is applied; for the 8-B / 4-B gathers of these kernels it is an upper estimate.
copy_fetch = fetch.get("__amd_rocclr_copyBuffer", {}).get("kb_per_step", 0.0) * 1024
copy_write = write.get("__amd_rocclr_copyBuffer", {}).get("kb_per_step", 0.0) * 1024
out = {
    "tag": tag, "steps_profiled": steps,
    "factor": {"fetch_size_kb": kb(fetch, factor_kernels), "write_size_kb": kb(write, factor_kernels)},
    "solve": {"fetch_size_kb": kb(fetch, solve_kernels), "write_size_kb": kb(write, solve_kernels)},
    "calibration": {"copyBuffer_fetch_bytes": copy_fetch, "copyBuffer_write_bytes": copy_write,
                    "note": "copyBuffer moves the same bytes in and out: fetch/write ratio = %.3f (guide: 0.5)"
                            % (copy_fetch / copy_write if copy_write else 0.0)},
}
out["prologue"] = {"fetch_size_kb": kb(fetch, ["k_prologue"]), "write_size_kb": kb(write, ["k_prologue"]),
                   "note": "one launch before every (re)factorisation: copy of the caller's values (twice with a bottom forest: "
                           "original order and forest order), zeros of the big-front buffers, status word; not in `factor`, as in round 2"}
out["factor"]["hbm_bytes_corrected"] = (2 * out["factor"]["fetch_size_kb"] + out["factor"]["write_size_kb"]) * 1024
out["solve"]["hbm_bytes_corrected"] = (2 * out["solve"]["fetch_size_kb"] + out["solve"]["write_size_kb"]) * 1024
json.dump({"fetch": fetch, "write": write}, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
