"""Per-layer table of the convolutions of one FF-PWC forward (1 pair 448x1024): every ops.conv2d call recorded, each
distinct shape replayed in isolation (time, useful TFLOP/s, grid size)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from focusflow_official_amd import ops
from focusflow_official_amd.pwcnet import FF_PWCNET

calls = collections.OrderedDict()
orig = ops.conv2d


def rec(xs, wpack, bias, cout, kh, kw, stride=1, pad=(0, 0), **kw_):
    out = orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
    key = (tuple(tuple(x.shape) for x in xs), cout, kh, kw, stride, kw_.get("dilation", 1))
    if key not in calls:
        calls[key] = dict(n=0, args=(xs, wpack, bias, cout, kh, kw, stride, pad, dict(kw_)))
    calls[key]["n"] += 1
    return out


cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
torch.manual_seed(0)
m = FF_PWCNET(cfg).cuda().eval()
with torch.no_grad():
    m.netExtractor.netOne[0].weight.mul_(1 / 255.0); m.netExtractor.mask_netOne[0].weight.mul_(1 / 255.0)
g = torch.Generator().manual_seed(0)
i1 = torch.randint(0, 256, (1, 3, 448, 1024), generator=g).float().cuda()
i2 = torch.roll(i1, (3, -5), (2, 3))
m1 = ((torch.rand(1, 1, 448, 1024, generator=g) < 2000 / (448 * 1024)).float() * 255).cuda()
with torch.no_grad():
    m(i1, i2, m1, m1, test_mode=True)
    ops.conv2d = rec
    import focusflow_official_amd.pwcnet as pw
    m(i1, i2, m1, m1, test_mode=True)
    ops.conv2d = orig
rows = []
for key, c in calls.items():
    xs, wpack, bias, cout, kh, kw, stride, pad, kw_ = c["args"]
    kw_ = {k: v for k, v in kw_.items() if k != "out"}
    for _ in range(2):
        out = orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    cin = sum(x.shape[3] for x in xs)
    npx = out.shape[0] * out.shape[1] * out.shape[2]
    fl = 2.0 * npx * cout * cin * kh * kw
    rows.append((us * c["n"], us, c["n"], key, fl / us / 1e6))
tot = sum(r[0] for r in rows)
print(f"total conv time (isolated replay) {tot / 1e3:.2f} ms per forward, {sum(r[2] for r in rows)} launches")
for t, us, n, key, tf in sorted(rows, reverse=True)[:45]:
    shp = "+".join(str(s[3]) for s in key[0])
    print(f"{t / 1e3:6.3f} ms {100 * t / tot:5.1f}%  n={n:2d} {us:7.1f} us  {tf:6.1f} TF/s  {key[0][0][1]}x{key[0][0][2]} cin {shp:>14} -> {key[1]:3d} k{key[2]}x{key[3]} s{key[4]} d{key[5]}")
