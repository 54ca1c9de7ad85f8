#!/usr/bin/env python3
"""Ordered kernel list of one steady-state step from a rocprofv3 --kernel-trace CSV (one-lane run): start offset, duration,
gap to the previous kernel's end, grid / workgroup size, kernel name.  usage: trace_sequence.py <kernel_trace.csv> [adam launches per step, default 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
NA = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() and int(sys.argv[2]) <= 8 else 2
s, e = idx[-5 - NA] + 1, idx[-5] + 1
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:56]
t0 = int(rows[s]['Start_Timestamp']); prev = t0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    w = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    print('%8.1f  dur %7.1f  gap %5.1f  wgs %6d x %4d  lds %6s vgpr %4s  %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3,
          g // max(w, 1), w, r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), short(r['Kernel_Name'])))
    prev = max(prev, en)
