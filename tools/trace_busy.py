"""GPU occupancy of one training (or forward) step from a rocprofv3 kernel trace: wall time of the step, the union of kernel
intervals (GPU busy), time with two or more kernels in flight, the largest idle gaps with the kernels either side, and the
kernels by summed duration inside the step.

    python tools/trace_busy.py <kernel_trace.csv> [anchor-kernel-substring] [step-index-from-end]

A step is the span between two consecutive launches of the anchor kernel (one launch per step: lookup_bwd_all_kernel in a
training step, corr_build_kernel in a forward)."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:name.index("(")] if "(" in name and not name.startswith("(") else name[:90]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "lookup_bwd_all_kernel"
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    marks = [s for s, e, n, q in rows if anchor in n]
    if len(marks) < back + 1:
        sys.exit(f"anchor {anchor!r}: {len(marks)} launches")
    t0, t1 = marks[-back - 1], marks[-back]
    step = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and s < t1]
    wall = (t1 - t0) / 1e3
    busy = over = 0
    gaps = []
    cur_end = t0
    prev = "(step start)"
    events = []
    for s, e, n, q in step:
        events.append((s, 1))
        events.append((e, -1))
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, prev, n))
        if e > cur_end:
            prev = n
            cur_end = e
    events.sort()
    depth, last = 0, t0
    for t, d in events:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    print(f"step {wall:.0f} us  kernels {len(step)}  busy {busy / 1e3:.0f} us ({busy / 10 / wall:.1f} %)  >=2 kernels in flight {over / 1e3:.0f} us  "
          f"queues {len({q for _, _, _, q in step})}")
    tot = sum(g for g, _, _ in gaps)
    print(f"idle {tot:.0f} us in {len(gaps)} gaps; gaps >= 20 us: {sum(g for g, _, _ in gaps if g >= 20):.0f} us; "
          f"gaps < 5 us: {sum(g for g, _, _ in gaps if g < 5):.0f} us in {sum(1 for g, _, _ in gaps if g < 5)}")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"  {g:8.1f} us   {short(a)[:60]:60s} -> {short(b)[:60]}")
    by = defaultdict(lambda: [0, 0])
    for s, e, n, q in step:
        by[short(n)][0] += e - s
        by[short(n)][1] += 1
    print("kernels by summed duration (us, launches):")
    for n, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:28]:
        print(f"  {d / 1e3:9.0f} {c:5d}  {n[:110]}")
    if len(sys.argv) > 5:          # timeline: <substring> <occurrence> [count] - launches from that one on
        sub, occ = sys.argv[4], int(sys.argv[5])
        cnt = int(sys.argv[6]) if len(sys.argv) > 6 else 80
        idx = [i for i, (s_, e_, n, q) in enumerate(step) if sub in n]
        if len(idx) > occ:
            i0 = idx[occ]
            base = step[i0][0]
            print(f"timeline from launch {occ} of {sub!r} (start us, duration us, queue, kernel):")
            for s_, e_, n, q in step[i0:i0 + cnt]:
                print(f"  {(s_ - base) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f}  q{q}  {short(n)[:100]}")
    byq = defaultdict(int)
    for s, e, n, q in step:
        byq[q] += e - s
    print("summed duration by queue:", {q: round(d / 1e3) for q, d in byq.items()})


if __name__ == "__main__":
    main()
