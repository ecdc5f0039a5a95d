"""64 -> 64 3x3 layer of the encoders (16 x 192 x 256, and the training crop 16 x 184 x 248): time and error against fp64 of
whatever route ff_conv2d_fwd takes (FF_CONV_WS64=0/1), plain / normalise-on-load + statistics / residual + BatchNorm fold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from focusflow_official_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(3)
def run(b, h, w, mode, check):
    cin = cout = 64
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24
    bias = torch.randn(cout, generator=g)
    rows = torch.empty(cout, 9 * cin, device=dev)
    ops.pack_conv_weight(wt.to(dev), rows, cin, 0)
    wp = ops.pack_split(rows); frag = ops.pack_frag16(wp, cout)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = {}
    ref = None
    if mode == "inorm+stats":
        sc, sh = torch.rand(b, cin, generator=g) + 0.5, torch.randn(b, cin, generator=g)
        kw = dict(in_scale=sc.to(dev), in_shift=sh.to(dev), in_act=1, want_stats=True)
        if check: ref = F.conv2d(torch.relu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]), wt.double(), bias.double(), padding=1)
    elif mode == "res+bn":
        res = torch.randn(b, cout, h, w, generator=g)
        cs, ct = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        kw = dict(act=1, ch_scale=cs.to(dev), ch_shift=ct.to(dev), res=res.permute(0, 2, 3, 1).contiguous().to(dev), act_res=1)
        if check: ref = torch.relu(torch.relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=1) * cs.double()[None, :, None, None] + ct.double()[None, :, None, None]) + res.double())
    else:
        if check: ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    f = lambda: ops.conv2d([xn], wp, bias.to(dev), cout, 3, 3, 1, (1, 1), w_fmt=1, w_frag=frag, **kw)
    for _ in range(3): y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): y = f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    st = None
    if isinstance(y, tuple): y, st = y
    msg = f"{b}x{h}x{w} {mode:12s}: {us:7.1f} us  {2 * b * h * w * 64 * 576 / us / 1e6:6.1f} TF/s useful"
    if check:
        yc = y.permute(0, 3, 1, 2).cpu().double()
        msg += f"  max err {float((yc - ref).abs().max()):.2e}"
        if st is not None:
            s = st.cpu()
            want = torch.stack([ref.sum((2, 3)), (ref * ref).sum((2, 3))], -1)
            msg += f"  stats rel err {float(((s - want).abs() / want.abs().clamp_min(1)).max()):.2e}"
    print(msg, flush=True)
for mode in ("plain", "inorm+stats", "res+bn"):
    run(2, 96, 128, mode, True) if False else None
    run(8, 72, 120, mode, True)
    run(16, 192, 256, mode, False)
    run(16, 184, 248, mode, False)
