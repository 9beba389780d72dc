#!/usr/bin/env python3
"""Do the kernels of different rooms overlap?  From a rocprofv3 --kernel-trace of tools/rooms_native: over the steady part of the
run (the middle 60 % of the trace), the share of wall time with 0, 1, 2, ... kernels running at once, the share of time each
kernel family is running, and the mean number of streams (= rooms' queues) with a kernel in flight.   usage: rooms_overlap.py <out dir>"""
import csv, glob, sys
rows = list(csv.DictReader(open(sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', r.get('Stream_Id', '0'))) for r in rows]
icp = [e for e in ev if e[2].startswith("k_icp_iter")]   # (the run proper: the contexts are created, and their frames rendered, long before)
t0, t1 = min(e[0] for e in icp), max(e[1] for e in icp)
a, b = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
ev = [e for e in ev if e[1] > a and e[0] < b]
pts = []
for s, e, name, q in ev:
    pts.append((max(s, a), 1))
    pts.append((min(e, b), -1))
pts.sort()
hist, cur, last = {}, 0, a
for t, d in pts:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
hist[cur] = hist.get(cur, 0) + (b - last)
span = b - a
print("kernels running at once (share of wall time over the middle 60 % of the run):")
for k in sorted(hist):
    if hist[k] / span >= 0.002:
        print(f"  {k:2d}: {hist[k] / span:6.3f}")
print(f"  mean concurrency {sum(k * v for k, v in hist.items()) / span:.2f}; GPU idle {hist.get(0, 0) / span:.3f}")
fam = {}
for s, e, name, q in ev:
    key = name.split('<')[0]
    fam[key] = fam.get(key, 0) + (min(e, b) - max(s, a))
print("kernel-time per family / wall time (can exceed 1 when several rooms run the same family at once):")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {k:28s} {v / span:6.3f}")
