#!/usr/bin/env python3
"""Per-kernel HBM traffic of one config-4 solve from the two counter passes of tools/pmc_cfg4.sh.
    python tools/pmc_cfg4_summary.py gpurun_out/<tag> [solves in the run = reps + 3 warm-ups]
FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE under-reports streaming reads by half on gfx950 (MI355X_MICROARCH.md, HBM
section; calibrated in tools/summarize_profile.py on the copy kernels): corrected bytes = 2 * fetch + write."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]; solves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, col in (("pmc_fetch", 0), ("pmc_write", 1)):
    f = max(glob.glob(os.path.join(src, which, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs3::", "")
        tot[n][col] += float(r["Counter_Value"]) * 1024
        if col == 0: tot[n][2] += 1
out = {}
for n, (fe, wr, calls) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    if not (n.startswith("k_") or "copyBuffer" in n): continue
    out[n] = {"launches_per_solve": calls / solves, "fetch_MB_per_solve": fe / solves / 1e6, "write_MB_per_solve": wr / solves / 1e6,
              "hbm_MB_corrected_per_solve": (2 * fe + wr) / solves / 1e6}
print(json.dumps(out, indent=1))
