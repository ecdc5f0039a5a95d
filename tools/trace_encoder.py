"""The encoder phase of one forward out of a rocprofv3 --kernel-trace CSV: everything between the last kernel of the previous
forward's update loop and the first lookup of forward number FWD (default 4) in the trace.  Per queue (= HIP stream): busy time and
kernels; over all queues: the span, the time at least one kernel runs, the time two or more run, and the kernels in
start order (FULL=1).
    python tools/trace_encoder.py gpurun_out/<tag>_kernel_trace.csv"""
import collections
import csv
import os
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
idx = [i for i, e in enumerate(ev) if "lookup_dma_kernel" in e[2]]
if len(idx) < 14:
    sys.exit("fewer than two forwards in the trace")
fwd = int(os.environ.get("FWD", "4"))      # which forward of the trace (0-based; bench.py --steps 4 --warmup 2: 2..5 are the timed steps)
first = idx[12 * fwd]           # first lookup of that forward (12 iterations)
prev_last = idx[12 * fwd - 1]   # last lookup of the forward before
lo = max(i for i in range(prev_last, first) if "mask_upsample" in ev[i][2]) + 1
seg = ev[lo:first]


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]


t0, t1 = seg[0][0], ev[first][0]
print(f"encoder phase: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us (first kernel start -> first lookup start)")
pts = []
for s, e, n, q in seg:
    pts += [(s, 1), (e, -1)]
pts.sort()
lvl, last, t_any, t_multi = 0, t0, 0, 0
for t, d in pts:
    if lvl >= 1:
        t_any += t - last
    if lvl >= 2:
        t_multi += t - last
    lvl += d
    last = t
print(f"at least one kernel running {t_any / 1e3:.1f} us, two or more {t_multi / 1e3:.1f} us, none {(t1 - t0 - t_any) / 1e3:.1f} us; summed kernel time {sum(e - s for s, e, _, _ in seg) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for s, e, n, q in seg:
    byq[q].append((s, e, n))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks)
    print(f"queue {q}: {len(ks)} launches, busy {busy / 1e3:.1f} us, from {(ks[0][0] - t0) / 1e3:.1f} to {(max(e for _, e, _ in ks) - t0) / 1e3:.1f} us")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ks:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"      {t / 1e3:8.1f} us  n={c:3d}  {n}")
if os.environ.get("FULL"):
    for s, e, n, q in seg:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q[-2:]} {short(n)}")
