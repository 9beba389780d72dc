#!/usr/bin/env python3
"""Print a slice of the kernel + memcpy timeline of a rocprofv3 trace dir (relative start, duration, queue, name)."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + '/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'Q' + r.get('Queue_Id', '?'), r['Kernel_Name'][:34]))
for f in glob.glob(d + '/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy', r.get('Direction', r.get('Name', 'memcpy'))[:34]))
ev.sort()
i0 = int(len(ev) * float(sys.argv[3]) if len(sys.argv) > 3 else len(ev) * 0.6)
t0 = ev[i0][0]
for s, e, q, n in ev[i0:i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 70]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {q:6s} {n}")
