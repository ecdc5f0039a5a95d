"""Per-layer table of the forward convolutions: record every ops.conv2d call - and every ops.gru_pass call, two
convolutions in one launch - of one FF-RAFT step (B=8, 384x512, 12 iterations), replay each distinct shape in isolation and
print time, useful TFLOP/s and minimum-traffic GB/s, sorted by their share of the step."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from focusflow_official_amd import FF_RAFT_FUSION, ops

calls = collections.OrderedDict()
orig = ops.conv2d


def rec(xs, wpack, bias, cout, kh, kw, stride=1, pad=(0, 0), **kw_):
    out = orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
    key = (tuple(tuple(x.shape) for x in xs), cout, kh, kw, stride, pad if isinstance(pad, tuple) else (pad, pad),
           kw_.get("w_fmt", 0), kw_.get("res") is not None)
    if key not in calls:
        calls[key] = dict(n=0, args=(xs, wpack, bias, cout, kh, kw, stride, pad, dict(kw_)))
    calls[key]["n"] += 1
    return out


cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg).cuda().eval()
b = int(os.environ.get("B", 8))
g = torch.Generator().manual_seed(0)
im = [torch.randint(0, 256, (b, 3, 384, 512), generator=g).float().cuda() for _ in range(2)]
mk = [((torch.rand(b, 1, 384, 512, generator=g) < 0.0025).float() * 255).cuda() for _ in range(2)]
import focusflow_official_amd.cce as cce, focusflow_official_amd.fn as fn, focusflow_official_amd.update_block as ub
orig_gp = ops.gru_pass
gp_calls = collections.OrderedDict()


def rec_gp(direction, *a):
    out = orig_gp(direction, *a)
    gp_calls.setdefault(direction, dict(n=0, args=(direction,) + a))["n"] += 1
    return out


orig_fp = ops.fusion_pair
fp_calls = collections.OrderedDict()


def rec_fp(img, mask, *a):
    lazy = isinstance(img, ops.LazyAct)
    key = (tuple(img.shape), lazy, lazy and img.res is not None)
    if key not in fp_calls:       # (a lazy input is overwritten by the launch: keep copies of the raw tensors for the replay)
        keep = [ops.LazyAct(v.t.clone(), v.stats, v.count, v.eps, v.act, v.res) if isinstance(v, ops.LazyAct) else v.clone() for v in (img, mask)]
        fp_calls[key] = dict(n=0, args=(keep[0], keep[1]) + a)
    fp_calls[key]["n"] += 1
    return orig_fp(img, mask, *a)


ops.conv2d = rec
ops.gru_pass = rec_gp
ops.fusion_pair = rec_fp
with torch.no_grad():
    m(im[0], im[1], mk[0], mk[1], raft_iters=12, test_mode=True)
ops.conv2d = orig
ops.gru_pass = orig_gp
ops.fusion_pair = orig_fp


def numel(x):
    return x.t.numel() if isinstance(x, ops.SplitT) else x.numel()


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
    if isinstance(out, tuple):      # want_stats: (output, statistics)
        out = out[0]
    cin = sum(x.shape[3] for x in xs)
    npx = out.shape[0] * out.shape[1] * out.shape[2]
    fl = 2.0 * npx * cout * cin * kh * kw
    by = (sum(numel(x) for x in xs) + numel(out) * (2 if key[-1] else 1)) * 4
    rows.append((us * c["n"], us, c["n"], key, fl / us / 1e6, by / us / 1e3))
for direction, c in gp_calls.items():       # z|r (384 -> 256) and q (384 -> 128) of one pass, 1x5 or 5x1, one launch
    for _ in range(2):
        orig_gp(*c["args"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        orig_gp(*c["args"])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    h = c["args"][3]
    npx = h.shape[0] * h.shape[1] * h.shape[2]
    fl = 2.0 * npx * 384 * 384 * 5
    by = npx * (128 * 3 + 384 + 128 * 2) * 4        # h, its split copy, the motion features, the context shares of the gates; new state twice
    kh, kw = ((1, 5), (5, 1))[direction]
    rows.append((us * c["n"], us, c["n"], (((h.shape[0], h.shape[1], h.shape[2], 384),), 384, kh, kw, 1, (kh // 2, kw // 2), 1, False), fl / us / 1e6, by / us / 1e3))
for key, c in fp_calls.items():       # a bidirectional fusion unit: two C x C 1x1 convolutions (+ the normalisation in front of it when lazy)
    shape, lazy, has_res = key
    for _ in range(2):
        orig_fp(*c["args"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        orig_fp(*c["args"])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    b_, h_, w_, c_ = shape
    npx = b_ * h_ * w_
    fl = 2.0 * npx * 2 * c_ * c_
    by = npx * c_ * 4 * (2 + 2 + (2 if has_res else 0))
    rows.append((us * c["n"], us, c["n"], (((b_, h_, w_, 2 * c_),), 2 * c_, 1, 1, 1, (0, 0), 1, bool(lazy)), fl / us / 1e6, by / us / 1e3))
tot = sum(r[0] for r in rows)
print(f"total conv time (isolated replay) {tot / 1e3:.2f} ms per step")
for t, us, n, key, tf, gb in sorted(rows, reverse=True)[:40]:
    shp = "+".join(str(s[3]) for s in key[0])
    if key[1] == 384 and key[0][0][3] == 384:
        shp = "gru pass"
    if key[2] == 1 and key[3] == 1 and key[0][0][3] == key[1] and len(key[0]) == 1 and key[1] in (128, 192):
        shp = "fusion" + ("+norm" if key[7] else "")
    print(f"{t / 1e3:6.2f} ms {100 * t / tot:5.1f}%  n={n:3d} {us:7.1f} us  {tf:6.1f} TF/s {gb:6.0f} GB/s  {key[0][0][1]}x{key[0][0][2]} cin {shp:>11} -> {key[1]:3d} k{key[2]}x{key[3]} s{key[4]} res={int(key[7])}")
