#!/usr/bin/env python3
"""Timeline of the LAST fused factor+solve step in a rocprofv3 kernel trace (a k_prologue step that holds backward-sweep
launches and no stand-alone copy of the right-hand sides):  tools/fused_step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
pro = [i for i, n in enumerate(names) if "k_prologue" in n] + [len(rows)]
pick = None
for a, b in zip(pro[:-1], pro[1:]):
    seg = names[a:b]
    if any("k_bwd" in n for n in seg) and not any("copyBuffer" in n for n in seg):
        pick = (a, b)
a, b = pick
t0 = int(rows[a]["Start_Timestamp"])
qend = {}
for r in rows[a:b]:
    if "at::native" in r["Kernel_Name"]:
        break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    gap = (s - qend[q]) / 1e3 if q in qend else 0.0
    qend[q] = e
    print("%-36s q%s start %7.1f dur %5.1f gap %6.1f grid %s,%s,%s" % (r["Kernel_Name"].replace("void cs3::", "").replace("cs3::", "")[:36], q,
          (s - t0) / 1e3, (e - s) / 1e3, gap, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
