"""Copy the counter CSVs of tools/prof_pmc.sh keeping only the kernels DESIGN.md quotes and the first N dispatches of each
(the summaries under profiles/ are made from the full CSVs; the trimmed ones are the evidence that travels in git).
    python tools/trim_pmc.py <src dir> <dst dir> [N=36] kernel-substring ..."""
import collections, csv, glob, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 36
keys = [a for a in sys.argv[3:] if not a.isdigit()]
os.makedirs(dst, exist_ok=True)
for f in sorted(glob.glob(os.path.join(src, "pmc*.csv"))):
    seen = collections.defaultdict(set)
    with open(f) as fi, open(os.path.join(dst, os.path.basename(f)), "w", newline="") as fo:
        rd = csv.DictReader(fi)
        wr = csv.DictWriter(fo, fieldnames=rd.fieldnames)
        wr.writeheader()
        for r in rd:
            k = r["Kernel_Name"]
            if not any(s in k for s in keys):
                continue
            d = seen[k]
            if r["Dispatch_Id"] not in d and len(d) >= n:
                continue
            d.add(r["Dispatch_Id"])
            wr.writerow(r)
if os.path.exists(os.path.join(src, "kernel_stats.csv")):
    shutil.copy(os.path.join(src, "kernel_stats.csv"), dst)
