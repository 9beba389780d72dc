#!/usr/bin/env python3
"""Kernels, copies and GPU idle gaps longer than 5 ms in a rocprofv3 --kernel-trace --memory-copy-trace run, with
the neighbouring timeline rows.  usage: tools/longgap.py <dir written by tools/trace_run.sh>"""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + '/*/*_kernel_trace.csv'))[-1]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(f))))
mc = []
for g in glob.glob(d + '/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(g)):
        mc.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size',''))))
allv = sorted(ev + mc)
t0 = allv[0][0]
print("long kernels/copies (>5 ms):")
for s, e, n in allv:
    if e - s > 5e6: print(f"  {(s-t0)/1e6:10.2f} ms  dur {(e-s)/1e6:8.2f} ms  {n}")
print("long gaps (>5 ms) in the last 30 % of the run:")
busy = allv[0][1]
cut = allv[int(len(allv) * 0.7)][0]
for i in range(1, len(allv)):
    s, e, n = allv[i]
    if s - busy > 5e6 and s > cut:
        print(f"  gap {(s-busy)/1e6:8.2f} ms at {(busy-t0)/1e6:10.2f} ms: after {allv[i-1][2]} -> before {n}")
        for j in range(max(0, i - 6), min(len(allv), i + 6)):
            print(f"       {(allv[j][0]-t0)/1e6:10.3f} {(allv[j][1]-allv[j][0])/1e3:9.1f} us {allv[j][2]}")
    busy = max(busy, e)
