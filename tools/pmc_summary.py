#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2:] or None
for k, d in sorted(acc.items()):
    if want and not any(w in k for w in want):
        continue
    n = max(len(v) for v in d.values())
    print(f"== {k}  (dispatches per pass ~{n})")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean={sum(v)/len(v):16.1f}  n={len(v)}")
