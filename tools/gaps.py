#!/usr/bin/env python3
"""Idle time of the GPU between consecutive kernels of a rocprofv3 --kernel-trace run, grouped by the pair
(previous kernel -> next kernel).  usage: tools/gaps.py <dir with *_kernel_trace.csv> [first_fraction_to_skip]"""
import csv, glob, sys
from collections import defaultdict
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in rows))
skip = int(len(ev) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.3))
ev = ev[skip:]
gaps = defaultdict(lambda: [0, 0])
busy_end = ev[0][1]
total_gap = 0
for (s, e, n), prev in zip(ev[1:], ev[:-1]):
    g = s - busy_end
    if g > 0:
        k = (prev[2], n)
        gaps[k][0] += g
        gaps[k][1] += 1
        total_gap += g
    busy_end = max(busy_end, e)
span = ev[-1][1] - ev[0][0]
print(f"span {span/1e3:.0f} us, idle {total_gap/1e3:.0f} us ({100*total_gap/span:.1f} %), kernels {len(ev)}")
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{g/1e3:9.0f} us  n={c:5d}  avg {g/c/1e3:6.2f} us   {k[0]}  ->  {k[1]}")
