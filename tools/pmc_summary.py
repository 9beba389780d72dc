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

# traffic file for bench.py's roofline.traffic: the integrate stage = pass A (k_integrate<false>) + pass B
# (k_integrate_detail3<false>), one dispatch of each per frame
import json, os
fetch = write = 0.0
parts = []
for k, d in acc.items():
    if (k.startswith("void k_integrate<false") or k.startswith("void k_integrate_detail3<false")) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 * 2   # gfx950: FETCH_SIZE reads 1/2 of wide streams
        w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
        fetch += f
        write += w
        parts.append({"kernel": k, "fetch_bytes_corrected_x2": int(f), "write_bytes": int(w)})
if parts:
    out = {"kernels": parts, "volume": int(os.environ.get("HSK_PMC_VOLUME", "512")), "bytes_per_launch": int(fetch + write),
           "fetch_bytes_corrected_x2": int(fetch), "write_bytes": int(write),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of `python bench.py`), tools/pmc.sh; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request on 16-B/lane streams)"}
    with open(os.path.join(root, "integrate_traffic.json"), "w") as f:
        json.dump(out, f)
    print("traffic:", out)
