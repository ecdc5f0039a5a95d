"""Error of a stride-1 convolution against fp64 (CPU), plain inputs and gradient-like inputs scaled through x_amax."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from focusflow_official_amd import ops
dev = "cuda"
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(dev)
def run(b, h, w, cin, cout, kh, kw, kind):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, cin, h, w, generator=g)
    if kind == "grad":
        x = x * 1e-6 * torch.exp(3 * torch.randn(b, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (kh * kw * cin) ** 0.5
    rows = torch.empty(cout, kh * kw * cin, device=dev)
    ops.pack_conv_weight(wt.to(dev), rows, cin, 0)
    wp = ops.pack_split(rows)
    frag = ops.pack_frag16(wp, cout)
    xn = nhwc(x)
    amax = None
    if kind == "grad":
        _, amax = ops.act_bwd(xn, None, 0, 1.0, cin, want_amax=True)
    ref = F.conv2d(x.double(), wt.double(), padding=(kh // 2, kw // 2))
    out = {}
    for name, fr in (("rows", None), ("frag", frag)):
        y = ops.conv2d([xn], wp, None, cout, kh, kw, 1, (kh // 2, kw // 2), w_fmt=1, x_amax=amax, w_frag=fr)
        e = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs()
        out[name] = (float(e.max() / ref.abs().max()), float(e.mean() / ref.abs().mean()))
    print(f"{kind:5s} {b}x{h}x{w} {cin}->{cout} k{kh}x{kw}: " + "  ".join(f"{k}: max {v[0]:.2e} mean {v[1]:.2e}" for k, v in out.items()))
for kind in ("plain", "grad"):
    run(8, 46, 62, 128, 128, 3, 3, kind)
    run(8, 46, 62, 256, 256, 5, 1, kind)
    run(4, 92, 124, 96, 96, 3, 3, kind)
    run(2, 184, 248, 64, 64, 3, 3, kind)
    run(2, 46, 62, 128, 128, 3, 3, kind)
    run(2, 40, 48, 64, 64, 3, 3, kind)
    run(2, 23, 31, 256, 256, 1, 5, kind)
    run(1, 20, 30, 96, 96, 3, 3, kind)
