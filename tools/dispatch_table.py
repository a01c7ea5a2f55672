#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 --kernel-trace CSV: one row per (kernel, grid, workgroup, VGPR, LDS) with call count
and average / min duration.   dispatch_table.py <kernel_trace.csv> [min total us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
agg = defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').split('(')[0]
    key = (name[:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')),
           r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('SGPR_Count', ''), r.get('LDS_Block_Size', ''))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
print('%9s %5s %8s %8s  grid/wg vgpr/agpr/sgpr lds  kernel' % ('total_us', 'n', 'avg_us', 'min_us'))
for key, v in out:
    if sum(v) < floor:
        continue
    print('%9.0f %5d %8.1f %8.1f  %s/%s %s/%s/%s %s  %s' % (sum(v), len(v), sum(v) / len(v), min(v), key[1], key[2], key[3], key[4],
                                                         key[5], key[6], key[0]))
