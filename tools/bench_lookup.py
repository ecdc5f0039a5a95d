"""Micro-benchmark of ff_corr_lookup_fwd at BASELINE config-2 shapes (B=8, 48x64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from focusflow_official_amd import ops

b, h, w = int(os.environ.get("B", 8)), 48, 64
g = torch.Generator().manual_seed(0)
f1 = torch.randn(b, h, w, 256, generator=g).cuda()
f2 = torch.randn(b, h, w, 256, generator=g).cuda()
pyr = ops.corr_pyramid(ops.corr_volume(f1, f2), h, w)
coords = ops.coords_init(b, h, w, f1)
coords += (torch.rand(coords.shape, generator=g) * 16 - 8).cuda()   # test plumbing only
for name, spoil in (("cache-warm (back-to-back)", False), ("cache-cold (512 MB written between launches)", True)):
    junk = torch.empty(128 * 1024 * 1024, device="cuda") if spoil else None
    ts = []
    for i in range(30):
        if spoil:
            junk.fill_(float(i))
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = ops.corr_lookup(pyr, coords, 4); e.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
    ts = sorted(ts[5:])
    us = ts[len(ts) // 2]
    nbytes = 2904 * b * h * w
    print(f"{name}: median {us:.1f} us  min {ts[0]:.1f} us -> {nbytes / us / 1e3:.0f} GB/s algorithmic = {nbytes / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s")
