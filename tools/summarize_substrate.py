#!/usr/bin/env python3
"""Kernel durations of tools/bench_substrate.py from its rocprofv3 kernel trace -> profiles/<tag>_substrate_kernels.json
with the algorithmic bytes of each operation (12 B per CSC entry read or written, 4 B per column pointer, 8 B per vector
entry) and the rate they imply.   python tools/summarize_substrate.py <trace dir> <tag>"""
import csv, glob, json, os, sys, collections
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "*", "*kernel_trace.csv"))[0])))
by = collections.defaultdict(list)
for r in rows:
    if "cs3::" in r["Kernel_Name"]:
        by[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs3::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n, nnz = 50000, 501330
med = {k: sorted(v)[len(v) // 2] for k, v in by.items()}
runs = 4                                                        # every operation is called 1 + 3 times
ops = {   # operation -> ({kernel: launches per call}, algorithmic bytes: 12 B per CSC entry read or written, 4 B per pointer, 8 B per vector entry)
 "csc_mat_vec_ff": ({"k_matvec_rows": 1}, 12 * nnz + 4 * (n + 1) + 16 * n),
 "csc_norm": ({"k_col_abs_sums": 1, "k_max_reduce": 1}, 8 * nnz + 4 * (n + 1)),
 "csc_transpose / csc_to_csr": ({"k_histogram": 1, "k_scan": 1, "k_bucket_fill": 1, "k_bucket_sort": 1, "k_expand_columns": 1, "k_gather_pairs": 1}, 2 * 12 * nnz + 8 * (n + 1)),
 "coo_to_csc": ({"k_histogram": 1, "k_scan": 1, "k_bucket_fill": 1, "k_bucket_sort": 1, "k_gather_pairs": 1}, 16 * nnz + 12 * nnz + 4 * (n + 1)),
 "csc_add_ff (A + A')": ({"k_add_columns": 2, "k_scan": 1}, 2 * 12 * nnz + 12 * (2 * nnz - n) + 12 * (n + 1)),
 "csc_sub_matrix (2000 x 2000 selection)": ({"k_sub_matrix": 2, "k_scan": 1}, None),
 "find_islands (symmetric pattern)": ({"k_label_init": 1, "k_pattern_symmetric": 1, "k_label_hook": len(by.get("k_label_hook", [])) // runs, "k_label_jump": len(by.get("k_label_jump", [])) // runs}, 2 * (4 * nnz + 4 * (n + 1)) + 8 * n),
 "csc_stack_4_by_4_ff": ({"k_stack_4_by_4": 1}, 2 * 12 * nnz + 8 * (n + 1)),
}
out = {"what": "kernel time of every substrate entry point on the 50k Jacobian (501 330 nnz), rocprofv3 --kernel-trace of tools/bench_substrate.py; "
               "median duration per kernel x launches per call; the host-pointer API adds PCIe copies and allocations that are not in these numbers "
               "(the *_dev entry points avoid them)",
       "kernel_median_us": {k: round(v, 2) for k, v in sorted(med.items())}, "operations": {}}
for op, (ks, bytes_) in ops.items():
    us = sum(med.get(k, 0.0) * c for k, c in ks.items())
    out["operations"][op] = {"kernels": ks, "kernel_us_per_call": round(us, 1), "algorithmic_bytes": bytes_,
                             "GBs": round(bytes_ / us / 1e3, 1) if bytes_ and us else None}
json.dump(out, open(os.path.join(root, "profiles", tag + "_substrate_kernels.json"), "w"), indent=1)
print(json.dumps(out["operations"], indent=1))
