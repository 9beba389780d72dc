import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'))[-1]
n=int(sys.argv[2]) if len(sys.argv)>2 else 8
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:52]:52s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.1f}us min={float(r['MinNs'])/1000:8.1f} max={float(r['MaxNs'])/1000:8.1f} tot%={r['Percentage']}")
