"""Per-kernel time per step from a rocprofv3 kernel_stats.csv: tools/kstats.py <csv> <steps in the trace> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
k = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / n / 1e6:.3f} ms/step over {len(rows)} kernels")
for r in rows[:k]:
    print(f"{int(r['TotalDurationNs']) / n / 1e6:7.3f} ms/step  calls/step {int(r['Calls']) / n:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
