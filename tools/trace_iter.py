"""One update-loop iteration out of a rocprofv3 --kernel-trace CSV: kernels in start order with duration and the gap to the
previous kernel's end (all queues merged), for the LAST iteration of the LAST forward in the trace.
    python tools/trace_iter.py gpurun_out/<tag>_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
idx = [i for i, e in enumerate(ev) if "lookup_dma_kernel" in e[2]]
if len(idx) < 3:
    sys.exit("no lookups in the trace")
lo, hi = idx[-2], idx[-1]       # the last full iteration: from the second-to-last lookup up to the last one


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n[:78]


busy = 0
prev_end = ev[lo][0]
print(f"{'start us':>9} {'dur us':>7} {'gap us':>7}  kernel")
for s, e, n, q in ev[lo:hi]:
    print(f"{(s - ev[lo][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:7.1f}  q{q[-2:]} {short(n)}")
    busy += e - s
    prev_end = max(prev_end, e)
span = ev[hi][0] - ev[lo][0]
print(f"iteration span {span / 1e3:.1f} us, summed kernel time {busy / 1e3:.1f} us, launches {hi - lo}")
# whole forward: first kernel after the previous forward's last lookup ... crude: span between lookups 12 apart
if len(idx) >= 13:
    a, b = idx[-13], idx[-1]
    print(f"12 iterations: {(ev[b][0] - ev[a][0]) / 1e3:.1f} us")
