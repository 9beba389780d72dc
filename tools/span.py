#!/usr/bin/env python3
"""Per-frame span of the integrate kernels from a rocprofv3 --kernel-trace CSV: start of k_column_zrange to the last
end of pass A / pass B (which may overlap in the HSK_EXP_OVERLAP timing experiment).  usage: span.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
spans, cur = [], None
for s, e, name in ev:
    if name.startswith("k_column_zrange"):
        if cur: spans.append(cur)
        cur = [s, e]
    elif cur and ("k_integrate" in name):
        cur[1] = max(cur[1], e)
if cur: spans.append(cur)
d = [(b - a) / 1000.0 for a, b in spans][3:]
print("frames %d  integrate span mean %.1f us  min %.1f  max %.1f" % (len(d), sum(d) / len(d), min(d), max(d)))
