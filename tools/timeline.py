#!/usr/bin/env python3
"""Print one sweep's kernel timeline from a rocprofv3 --kernel-trace CSV (development tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = [r for r in csv.DictReader(open(f)) if 'at::native' not in r['Kernel_Name'] and 'rocprim' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_band' in r['Kernel_Name']]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    import re
    m = re.search(r'(k_\w+|nccl\w*|rccl\w*)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:40]
    print('%8.1f %8.1f dur %7.1f  q=%s %s' % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
