#!/usr/bin/env python3
"""Print the kernel timeline of the LAST factor(+solve) step found in a rocprofv3 kernel trace:
   tools/last_step_timeline.py <kernel_trace.csv> [anchor-substring, default k_prologue]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_prologue"
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if anchor in n)
t0 = int(rows[last]["Start_Timestamp"])
end = 0
for r in rows[last:]:
    if "at::native" in r["Kernel_Name"]:
        break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, e)
    print("%-46s q%s start %8.1f dur %7.1f grid %s,%s,%s" % (r["Kernel_Name"].replace("void cs3::", "")[:46], r["Queue_Id"], (s - t0) / 1e3,
                                                          (e - s) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
print("span %.1f us" % ((end - t0) / 1e3))
