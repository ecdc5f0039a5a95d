import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from focusflow_official_amd import ops
dev = "cuda"
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(dev)
def run(b, h, w, cins, cout, kh, kw, use_res, act):
    g = torch.Generator().manual_seed(1)
    cin = sum(cins)
    xs = [torch.randn(b, c, h, w, generator=g) for c in cins]
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (kh * kw * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(b, cout, h, w, generator=g)
    rows = torch.empty(cout, kh * kw * cin, device=dev)
    ops.pack_conv_weight(wt.to(dev), rows, cin, 0)
    wp = ops.pack_split(rows)
    frag = ops.pack_frag16(wp, cout)
    ref = F.conv2d(torch.cat(xs, 1).double(), wt.double(), bias.double(), padding=(kh // 2, kw // 2))
    if use_res: ref = ref + res.double()
    if act == 2: ref = torch.sigmoid(ref)
    if act == 3: ref = torch.tanh(ref)
    kw_ = dict(res=nhwc(res), act_res=act) if use_res else dict(act=act)
    y = ops.conv2d([nhwc(x) for x in xs], wp, bias.to(dev), cout, kh, kw, 1, (kh // 2, kw // 2), w_fmt=1, w_frag=frag, **kw_)
    e = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs()
    print(f"{b}x{h}x{w} {cins}->{cout} k{kh}x{kw} res={use_res} act={act}: max {float(e.max()):.2e} mean {float(e.mean()):.2e} frac>1e-5 {float((e > 1e-5).double().mean()):.4f}")
for cins in ([128], [128, 128]):
    for use_res in (False, True):
        for act in (0, 2, 3):
            run(2, 16, 16, cins, 128, 1, 5, use_res, act)
run(2, 16, 16, [128, 128], 256, 5, 1, True, 2)
run(8, 46, 62, [128, 128], 256, 5, 1, True, 2)
