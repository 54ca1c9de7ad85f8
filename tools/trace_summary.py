#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-step kernel time by kernel family, idle gaps, slowest GEMM launches.
usage: trace_summary.py <kernel_trace.csv> [--steps-from-end N]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'adam_kernel' in n]
import os
NA = int(os.environ.get('ADAMS_PER_STEP', '2'))
s, e = idx[-5 - NA] + 1, idx[-5] + 1            # a full steady-state step: two Adam launches; the last steps lack the next-batch prefetch
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]
tot = collections.defaultdict(lambda: [0, 0.0])
gaps = []
prev = int(rows[s]['Start_Timestamp'])
t0 = prev
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = short(r['Kernel_Name'])
    tot[k][0] += 1; tot[k][1] += (en - st) / 1e3
    if st - prev > 3000: gaps.append(((st - prev) / 1e3, (st - t0) / 1e3, k))
    prev = max(prev, en)
wall = (prev - t0) / 1e3
ksum = sum(v[1] for v in tot.values())
print('step wall %.0f us, kernels %.0f us, idle %.0f us' % (wall, ksum, wall - ksum))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print('  %-62s n=%3d  %8.1f us' % (k, v[0], v[1]))
print('gaps > 3us:', ' '.join('%.0f@%.0f(%s)' % (g, t, k[:14]) for g, t, k in sorted(gaps, reverse=True)[:14]))
