"""ff_act_bwd on the gradient shapes of the training step (8 pairs 368x496: 22 816 pixels at 1/8, 64-256 channels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focusflow_official_amd import ops
g = torch.Generator().manual_seed(0)
for (b, h, w, c, act) in [(8, 46, 62, 256, 1), (8, 46, 62, 128, 3), (8, 46, 62, 512, 1), (16, 184, 248, 64, 0), (8, 46, 62, 576, 0)]:
    dy = (torch.randn(b, h, w, c, generator=g) * 1e-4).cuda()
    y = torch.relu(torch.randn(b, h, w, c, generator=g)).cuda()
    for _ in range(3): gg, am = ops.act_bwd(dy, y if act else None, act, 1.0 if act else 0.25, c, want_amax=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gg, am = ops.act_bwd(dy, y if act else None, act, 1.0 if act else 0.25, c, want_amax=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = dy.numel() * 4 * (3 if act else 2)
    print(f"{b}x{h}x{w}x{c} act={act}: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s")
