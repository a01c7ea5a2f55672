#!/usr/bin/env python
"""Launch-by-launch timeline of ONE step from a rocprofv3 --kernel-trace CSV:  step_timeline.py <kernel_trace.csv> [out.txt]
The last complete step is cut at the last two `adam_kernel` dispatches; every dispatch between them is listed in
start order with its duration and the gap to the previous kernel's end (graph replay: one queue)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
for r in rows:
    r['_s'] = int(r['Start_Timestamp']); r['_e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['_s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[ks]]
if len(adam) < 2:
    sys.exit('fewer than two adam_kernel dispatches in the trace')
seg = rows[adam[-2] + 1:adam[-1] + 1]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
prev = rows[adam[-2]]['_e']
busy = 0
for r in seg:
    n = r[ks].replace('void ', '').replace('eve::', '').split('(')[0][:80]
    d = (r['_e'] - r['_s']) / 1e3
    busy += d
    print('%9.1f us  gap %6.1f  %s' % (d, (r['_s'] - prev) / 1e3, n), file=out)
    prev = max(prev, r['_e'])
print('step: %d launches, kernels %.3f ms, wall %.3f ms' % (len(seg), busy / 1e3, (seg[-1]['_e'] - rows[adam[-2]]['_e']) / 1e6), file=out)
