"""Micro-benchmark of one convolution shape through the C ABI (default: FF-RAFT layer1 3x3 64->64 at 192x256xB8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from focusflow_official_amd import ops

b, h, w = int(os.environ.get("B", 8)), int(os.environ.get("H", 192)), int(os.environ.get("W", 256))
cin, cout, k = int(os.environ.get("CIN", 64)), int(os.environ.get("COUT", 64)), int(os.environ.get("K", 3))
fmt = {"fp32": 0, "f16x3": 1, "f16": 2}[os.environ.get("FMT", "f16x3")]
g = torch.Generator().manual_seed(0)
x = torch.randn(b, h, w, cin, generator=g).cuda()
wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
bias = torch.randn(cout, generator=g).cuda()
if os.environ.get("ZERO"):          # all-zero operands: the same instruction stream at a fraction of the matrix pipe's power
    x.zero_(); wt.zero_()
wp = torch.empty(cout, k * k * cin, device="cuda")
ops.pack_conv_weight(wt, wp, cin)
if fmt:
    wp = ops.pack_split(wp)
for _ in range(3):
    y = ops.conv2d([x], wp, bias, cout, k, k, 1, k // 2, act=1, w_fmt=fmt)
torch.cuda.synchronize()
n = int(os.environ.get("REPS", 20))
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    y = ops.conv2d([x], wp, bias, cout, k, k, 1, k // 2, act=1, w_fmt=fmt)
e.record()
torch.cuda.synchronize()
us = a.elapsed_time(e) * 1e3 / n
fl = 2.0 * b * h * w * cout * cin * k * k
print(f"conv {k}x{k} {cin}->{cout} @ {b}x{h}x{w} fmt={os.environ.get('FMT', 'f16x3')}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s (fp32-equivalent)  "
      f"min HBM {(x.numel() + y.numel()) * 4 / us / 1e3:.0f} GB/s")
