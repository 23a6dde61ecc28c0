#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV: per launch start (relative), duration and the gap to the
previous launch on the same queue / to the latest end of any earlier launch.  Also sums by kernel.
    python tools/trace_timeline.py <kernel_trace.csv> [--first KERNEL_SUBSTRING] [--skip N] [--fused] [--out file.json]
--fused: the last FUSED factor + solve step (bench.py's timed step) rather than the last step of the trace.
The step shown is the LAST one in the trace: it starts at the last launch whose name contains --first
(default k_prologue; --skip N: the N-th from last, and later matches stay inside the step) and runs to the end of the
trace or to the next non-cs3 kernel."""
import csv, json, sys

path = sys.argv[1]
first = "k_prologue"
out = None
skip = 0
fused = False
args = sys.argv[2:]
while args:
    a = args.pop(0)
    if a == "--first": first = args.pop(0)
    elif a == "--out": out = args.pop(0)
    elif a == "--skip": skip = int(args.pop(0))
    elif a == "--fused": fused = True
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs3::", "")
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if not starts:
    sys.exit("no launch matching %r" % first)
i0 = starts[-1 - skip]
if fused:            # the last step that holds backward-sweep launches and no stand-alone copy of the right-hand sides
    bounds = starts + [len(rows)]
    for a0, b0 in zip(bounds[:-1], bounds[1:]):
        seg = [r["Kernel_Name"] for r in rows[a0:b0]]
        if any("k_bwd" in n for n in seg) and not any("copyBuffer" in n for n in seg):
            i0 = a0
i1 = i0
allowed = skip
while i1 + 1 < len(rows) and ("cs3::" in rows[i1 + 1]["Kernel_Name"] or "__amd_rocclr" in rows[i1 + 1]["Kernel_Name"]):
    if first in rows[i1 + 1]["Kernel_Name"]:
        if allowed == 0: break
        allowed -= 1
    i1 += 1
step = rows[i0:i1 + 1]
t0 = int(step[0]["Start_Timestamp"])
latest_end = t0
per_queue_end = {}
lines, by = [], {}
busy = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    gap_q = (s - per_queue_end[q]) / 1e3 if q in per_queue_end else None
    gap_all = (s - latest_end) / 1e3
    lines.append({"kernel": name(r), "queue": q, "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3,
                  "gap_same_queue_us": gap_q, "gap_after_all_earlier_us": gap_all,
                  "grid": [int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])], "wg": int(r["Workgroup_Size_X"])})
    per_queue_end[q] = e
    latest_end = max(latest_end, e)
    k = by.setdefault(name(r), [0, 0.0])
    k[0] += 1; k[1] += (e - s) / 1e3
wall = (latest_end - t0) / 1e3
# time covered by at least one kernel
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: cov += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cov += cur_e - cur_s
summary = {"launches": len(step), "wall_us": wall, "sum_of_durations_us": sum(l["dur_us"] for l in lines),
           "covered_us": cov / 1e3, "idle_gaps_us": wall - cov / 1e3,
           "by_kernel": {k: {"launches": v[0], "us": v[1]} for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}}
for l in lines:
    print("%9.1f us  +%7.1f us  %-34s q%-3s gap(queue) %s gap(all) %6.1f  grid %s x %d" % (
        l["start_us"], l["dur_us"], l["kernel"][:34], l["queue"],
        ("%6.1f" % l["gap_same_queue_us"]) if l["gap_same_queue_us"] is not None else "   -  ", l["gap_after_all_earlier_us"], l["grid"], l["wg"]))
print(json.dumps(summary, indent=1))
if out:
    json.dump({"summary": summary, "launches": lines}, open(out, "w"), indent=1)
