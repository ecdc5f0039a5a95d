"""ff_norm_bwd (statistics pass + apply pass) on the encoder shapes of the training step, with and without the max|dx| word."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focusflow_official_amd import ops
g = torch.Generator().manual_seed(0)
for (b, h, w, c) in [(16, 184, 248, 64), (16, 92, 124, 96), (16, 46, 62, 128), (8, 184, 248, 64)]:
    x = torch.randn(b, h, w, c, generator=g).cuda()
    dy = (torch.randn(b, h, w, c, generator=g) * 1e-4).cuda()
    st = ops.norm_stats(x, per_sample=True)
    y = ops.norm_apply(x, st, True, 1e-5, act=1)
    for use_amax in (False, True):
        def run():
            am = torch.zeros(1, dtype=torch.int32, device="cuda") if use_amax else None
            return ops.norm_bwd(x, dy, y, st, True, False, 1e-5, None, None, True, False, amax=am)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        mb = x.numel() * 4 / 1e6
        print(f"{b}x{h}x{w}x{c} ({mb:.0f} MB) amax={int(use_amax)}: {us:7.1f} us both passes  ({(2 * 2 + 1 + 1) * mb / us / 1e0:.0f} GB/s if 6 tensor passes)".replace(" GB/s", " MB/us"))
