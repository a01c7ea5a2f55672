#!/usr/bin/env python
"""Per-step kernel time table from a rocprofv3 kernel_stats.csv:  kstats.py <csv> <steps in the trace> [min ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
tot = 0.0
for r in rows:
    t = int(r['TotalDurationNs']) / steps / 1e6
    tot += t
    n = r['Name'].replace('void ', '').split('(')[0][:72]
    if t > floor:
        print(f"{t:7.3f} ms  x{int(r['Calls']) / steps:5.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {n}")
print(f"total {tot:.3f} ms/step")
