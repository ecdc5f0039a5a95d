"""Summarise the CSVs of tools/prof_pmc.sh: per kernel (name substring filter), average counter value per dispatch."""
import csv, glob, os, sys, collections
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if flt in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(f), k)
        for c, v in cs.items():
            print(f"    {c:34s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if flt in r["Name"]:
            print("trace", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
