#!/usr/bin/env python3
"""Developer tool: reads a rocprofv3 --kernel-trace csv and prints the timeline of the last dispatches of one kernel --
start / end relative to the first one shown, duration, queue and stream -- plus how much consecutive dispatches overlap.
usage: overlap_timeline.py <dir-with-*_kernel_trace.csv> <kernel-name-prefix> [count]"""
import csv, glob, os, sys

d, pref = sys.argv[1], sys.argv[2]
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 12
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(pref):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
rows = rows[-cnt:]
t0 = rows[0][0]
print("%d dispatches of %s*; times in ms relative to the first one shown" % (len(rows), pref))
print("%4s %10s %10s %9s %7s %7s %12s %12s" % ("#", "start", "end", "duration", "queue", "stream", "gap_to_prev", "overlap_prev"))
prev_end = prev_start = None
starts = []
for i, (s, e, q, st) in enumerate(rows):
    gap = "" if prev_end is None else "%.3f" % ((s - prev_start) / 1e6)
    ov = "" if prev_end is None else "%.3f" % (max(0, prev_end - s) / 1e6)
    print("%4d %10.3f %10.3f %9.3f %7s %7s %12s %12s" % (i, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, gap, ov))
    prev_end, prev_start = e, s
    starts.append(s)
if len(rows) > 2:
    per = (rows[-1][1] - rows[0][1]) / 1e6 / (len(rows) - 1)
    print("mean period (end to end): %.3f ms; mean duration %.3f ms" % (per, sum(e - s for s, e, _, _ in rows) / 1e6 / len(rows)))
