"""Micro-benchmark of the normalisation passes at the encoder's shapes (fnet: both frames of 8 pairs = 16 images)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from focusflow_official_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


junk = torch.empty(96 * 1024 * 1024, device="cuda")
for (b, h, w, c) in [(16, 192, 256, 64), (16, 96, 128, 96), (16, 48, 64, 128), (8, 192, 256, 64)]:
    x = torch.randn(b, h, w, c, device="cuda")
    r = torch.randn(b, h, w, c, device="cuda")
    nbytes = x.numel() * 4
    ops.begin_forward(x.device)
    st = ops.norm_stats(x, True)

    def stats():
        ops.begin_forward(x.device)
        return ops.norm_stats(x, True)

    def stats_cold():
        junk.fill_(1.0)
        return stats()
    t_fill = timeit(lambda: junk.fill_(1.0))
    us = timeit(stats)
    usc = timeit(stats_cold) - t_fill
    y = torch.empty_like(x)
    ua = timeit(lambda: ops.norm_apply(x, st, True, 1e-5, act=1, res=r, out=y))
    ua2 = timeit(lambda: ops.norm_apply(x, st, True, 1e-5, act=1, out=y))
    print(f"{b}x{h}x{w}x{c} ({nbytes / 1e6:.0f} MB): stats {us:.1f} us ({nbytes / us / 1e6:.2f} TB/s; cold {usc:.1f} us), apply+res {ua:.1f} us "
          f"({3 * nbytes / ua / 1e6:.2f} TB/s), apply {ua2:.1f} us ({2 * nbytes / ua2 / 1e6:.2f} TB/s)")
