#!/usr/bin/env python3
"""Developer tool: the last `count` dispatches of a rocprofv3 kernel trace (csv) as a timeline -- queue, kernel, grid, start / end in us
from the first of them.  usage: kernel_timeline.py <dir-with-*_kernel_trace.csv> [count] [name-filter]"""
import csv, glob, os, sys
d, count = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
flt = sys.argv[3] if len(sys.argv) > 3 else ""
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("queue %-3s %-34s grid %8s x %-4s  %9.1f -> %9.1f us  (%8.1f)" % (r["Queue_Id"], r["Kernel_Name"].split("(")[0][:34], r["Grid_Size_X"], r["Workgroup_Size_X"], s, e, e - s))
