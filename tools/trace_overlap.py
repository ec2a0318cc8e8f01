#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 --kernel-trace csv, print a timeline of the last N kernel dispatches (start, end,
queue, name) and how much of the time two kernels were running at once."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f  q%-3s %s" % (s / 1e3, e / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:40]))
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
busy = [0, 0, 0, 0, 0]
depth, prev = 0, ev[0][0]
for t, d in ev:
    busy[min(depth, 4)] += t - prev
    depth += d
    prev = t
print("time with 0/1/2/3/4+ kernels running (us):", [round(b / 1e3, 1) for b in busy])
