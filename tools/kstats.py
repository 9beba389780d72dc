"""kernel stats of a rocprofv3 --kernel-trace --stats run: the stats file's means beside MEDIANS over the trace (the first frame
of a scan takes a 1 ms pass B into every mean).  usage: kstats.py <out dir> [top n]"""
import csv, glob, sys
import statistics
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
med = {}
tr = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))
if tr:
    per = {}
    for r in csv.DictReader(open(tr[-1])):
        per.setdefault(r['Kernel_Name'], []).append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    med = {k: statistics.median(v) for k, v in per.items()}
for r in list(csv.DictReader(open(f)))[:n]:
    m = med.get(r['Name'])
    print(f"{r['Name'][:52]:52s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.1f}us" + (f" median={m:8.1f}" if m is not None else "") +
          f" min={float(r['MinNs'])/1000:8.1f} max={float(r['MaxNs'])/1000:8.1f} tot%={r['Percentage']}")
