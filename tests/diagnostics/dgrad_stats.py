"""Debug: dynamic range of the gradient tensors that enter the input-gradient convolutions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from argparse import Namespace
import oracle.ffraft_ref as orc
from oracle.weights import det_tensor
from conftest import golden_spec
from focusflow_official_amd import FF_RAFT_FUSION, ops
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
det_sd = {k: det_tensor(k, s) for k, s, _ in golden_spec()}
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
m.load_state_dict(det_sd, strict=True)
m = m.cuda().train()
m.flow_net.freeze_bn()
orig = ops.conv2d
rows = []
BWD = [False]
def hook(xs, wpack, bias, cout, kh, kw, stride=1, pad=(0, 0), **kw_):
    if kw_.get("x_amax") is not None:
        g = xs[0]
        a = g.abs()
        mx = a.max().item()
        nz = a[a > 0]
        med = nz.median().item() if nz.numel() else 0
        l1 = a.sum().item()
        small = a[a < mx * 2.0 ** -25].sum().item()
        out = orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
        relerr = -1.0
        if kw_.get("w_fmt") == 1:
            hw = wpack.view(torch.float16).view(wpack.shape[0], -1, 2, 32).float()
            wf = ((hw[:, :, 0] + hw[:, :, 1]) / 16.0).reshape(wpack.shape[0], -1)[:, :kh * kw * sum(x.shape[3] for x in xs)].contiguous()
            kw2 = dict(kw_); kw2["w_fmt"] = 0; kw2["x_amax"] = None
            ref = orig(xs, wf, bias, cout, kh, kw, stride, pad, **kw2)
            relerr = ((out - ref).abs().max() / ref.abs().max()).item()
        rows.append((tuple(g.shape), cout, kh, kw, mx, med, relerr, kw_["x_amax"].view(torch.float32).item()))
        return out
    elif BWD[0]:
        print("BWD conv without amax: x", tuple(xs[0].shape), "->", cout, kh, kw, "fmt", kw_.get("w_fmt"), "max|x| %.2e" % xs[0].abs().max().item())
    return orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
import focusflow_official_amd.fn as fn
ops.conv2d = hook
inp = orc.shifted_pair(1, 128, 128, seed=9)
preds = m(*[t.cuda() for t in inp], raft_iters=2)
BWD[0] = True
preds[-1].abs().mean().backward()
for r in rows:
    print("g%-22s ->%4d k%dx%d  max %.2e amaxword %.2e  median/max 2^%.1f  REL ERR vs fp32 conv: %.2e" % (r[0], r[1], r[2], r[3], r[4], r[7], np.log2(r[5] / r[4] + 1e-300), r[6]))
