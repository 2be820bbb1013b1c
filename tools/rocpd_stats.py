#!/usr/bin/env python3
"""Kernel statistics (the `rocprofv3 --kernel-trace --stats` summary) from a rocprofv3 rocpd database:
rocpd_stats.py <results.db> <out.csv>.  ROCm 7.2's rocprofv3 writes SQLite by default; this is the per-kernel
Calls / TotalDurationNs / AverageNs / MinNs / MaxNs / Percentage table of its `kernels` view."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 2)])
for r in rows[:8]:
    print(r[0][:100], r[1], round(r[3]), round(100 * r[2] / tot, 1))
