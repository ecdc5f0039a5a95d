"""profiles/r02_lookup_traffic*.json from the rocprofv3 --pmc CSVs of tools/prof_pmc.sh (tools/bench_lookup.py, ONLY=lookup).
HBM bytes per launch as MI355X_MICROARCH.md prescribes for gfx950: reads = TCC_EA0_RDREQ x 128 B (FETCH_SIZE = RDREQ x 64 B
under-reports wide reads by 2x), writes = WRITE_SIZE (KB, exact for 16-byte streaming stores).
usage: python tools/make_traffic_json.py <pmc dir> <kernel substring> <queries per launch> <bytes per query> <out.json> [note]"""
import collections, csv, glob, json, os, sys

d, flt, queries, per_q, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
note = sys.argv[6] if len(sys.argv) > 6 else ""
acc = collections.defaultdict(list)
name = None
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    for r in csv.DictReader(open(f)):
        if flt in r.get("Kernel_Name", ""):
            name = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
dur = None
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if flt in r["Name"]:
            dur = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
read_b = avg["TCC_EA0_RDREQ_sum"] * 128
write_b = avg["WRITE_SIZE"] * 1024
rec = {
    "kernel": name, "workload": note, "queries_per_launch": queries,
    "counters_avg_per_launch": avg,
    "read_bytes_per_launch": read_b, "read_note": "TCC_EA0_RDREQ x 128 B; FETCH_SIZE (KB) = RDREQ x 64 B under-reports by 2x on gfx950 (MI355X_MICROARCH.md HBM section)",
    "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_query": per_q, "algorithmic_bytes_per_launch": per_q * queries,
    "traffic_over_algorithmic": (read_b + write_b) / (per_q * queries),
    "l2_hit_rate": avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]) if "TCC_HIT_sum" in avg else None,
    "kernel_trace": dur,
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("read_bytes_per_launch", "write_bytes_per_launch", "traffic_over_algorithmic", "l2_hit_rate", "kernel_trace")}))
