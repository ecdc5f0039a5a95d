"""GPU idle time inside the steady state of a traced run: tools/gpu_busy.py <kernel_trace.csv> <window ms from the end | marker>
(a marker = part of a kernel name that occurs once per step, e.g. lpnorm_cleanup for clip_grad_norm_: the window is then
the last complete step, between the last two bursts of that kernel)
Union of all kernel intervals (streams overlap) over the window, the gaps between them, and which kernels precede gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
try:
    win = float(sys.argv[2]) * 1e6
    t_end = max(k[1] for k in ks)
    ks = [k for k in ks if k[0] >= t_end - win]
except ValueError:
    name, _, which = sys.argv[2].partition(":")          # marker or marker:index (which burst opens the window; default: the last but one)
    marks = [k[0] for k in ks if name in k[2]]
    bursts = [m for i, m in enumerate(marks) if i == 0 or m - marks[i - 1] > 5e6]       # first of each burst (5 ms apart)
    i0 = int(which) if which else len(bursts) - 2
    a, b = bursts[i0], bursts[i0 + 1]
    ks = [k for k in ks if a <= k[0] < b]
t_end = max(k[1] for k in ks)
t0 = ks[0][0]
busy, cur_s, cur_e, last_name = 0, ks[0][0], ks[0][1], ks[0][2]
gaps = []
for s, e, n in ks[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
        last_name = n
    elif e > cur_e:
        cur_e = e
        last_name = n
busy += cur_e - cur_s
wall = t_end - t0
ksum = sum(e - s for s, e, _ in ks)
print(f"window {wall / 1e6:.2f} ms: union busy {busy / 1e6:.2f} ms ({100 * busy / wall:.1f} %), kernel sum {ksum / 1e6:.2f} ms, {len(ks)} kernels, {len(gaps)} gaps")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo / 1e3:6.0f}..{hi / 1e3:.0f} us: {len(g):6d}, {sum(g) / 1e6:7.2f} ms")
by = collections.Counter()
for g, a, b in gaps:
    by[a[:70]] += g
print("idle time by the kernel BEFORE the gap:")
for n, g in by.most_common(12):
    print(f"  {g / 1e6:7.2f} ms  {n}")
