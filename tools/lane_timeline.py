#!/usr/bin/env python3
"""Per-stream view of a rocprofv3 --kernel-trace CSV for one steady-state train step: busy time per queue/stream, the
union of busy intervals, and the kernel sequence with start offsets (to see which lane waits for which).
usage: lane_timeline.py <kernel_trace.csv> [n_rows_to_print] [adam launches to skip from the end (default 40: clear of the run's drain phase)]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lane_key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
s, e = idx[-skip - 5] + 1, idx[-skip - 1] + 1          # two steady-state steps (two Adam launches each)
seg = rows[s:e]
t0 = int(seg[0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:44]
busy = collections.defaultdict(float)
iv = []
for r in seg:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy[r[lane_key]] += (b - a) / 1e3
    iv.append((a, b))
iv.sort()
union = 0; cs, ce = iv[0]
for a, b in iv[1:]:
    if a > ce: union += ce - cs; cs, ce = a, b
    else: ce = max(ce, b)
union += ce - cs
wall = (max(b for a, b in iv) - t0) / 1e3
print('lanes by %s; 2 steps: wall %.0f us, union busy %.0f us, sum of kernels %.0f us' % (lane_key, wall, union / 1e3, sum(busy.values())))
for k, v in sorted(busy.items()): print('  lane %s: busy %.0f us' % (k, v))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in seg[:n]:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f %7.1f  lane %-4s %s' % ((a - t0) / 1e3, (b - a) / 1e3, r[lane_key], short(r['Kernel_Name'])))
