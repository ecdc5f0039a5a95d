"""Debug: relative gradient errors of the frozen-BN training step against CPU autograd through the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
if os.environ.get("FF_LIB"):
    import focusflow_official_amd._hip as _h
    _h.LIB_PATH = os.environ["FF_LIB"]
from argparse import Namespace
import oracle.ffraft_ref as orc
from oracle.weights import det_tensor
from conftest import golden_spec
from focusflow_official_amd import FF_RAFT_FUSION
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
det_sd = {k: det_tensor(k, s) for k, s, _ in golden_spec()}
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
m.load_state_dict(det_sd, strict=True)
m = m.cuda().train()
if os.environ.get("FREEZE", "1") == "1":
    m.flow_net.freeze_bn()
from focusflow_official_amd import ops
_orig = ops.conv2d
MODE = os.environ.get("EXACT", "")          # "dgrad": input-gradient convs on the fp32 path; "fwd": all other convs
def _hook(xs, wpack, bias, cout, kh, kw, stride=1, pad=(0, 0), **kw_):
    is_d = kw_.get("x_amax") is not None
    if kw_.get("w_fmt") == 1 and ((MODE in ("dgrad", "both") and is_d) or (MODE in ("fwd", "both") and not is_d)):
        hw = wpack.view(torch.float16).view(wpack.shape[0], -1, 2, 32).float()
        wf = ((hw[:, :, 0] + hw[:, :, 1]) / 16.0).reshape(wpack.shape[0], -1)[:, :kh * kw * sum(x.shape[3] for x in xs)].contiguous()
        kw2 = dict(kw_); kw2["w_fmt"] = 0; kw2["x_amax"] = None
        return _orig(xs, wf, bias, cout, kh, kw, stride, pad, **kw2)
    return _orig(xs, wpack, bias, cout, kh, kw, stride, pad, **kw_)
if MODE:
    ops.conv2d = _hook
inp = orc.shifted_pair(1, 128, 128, seed=9)
preds = m(*[t.cuda() for t in inp], raft_iters=2)
preds[-1].abs().mean().backward()
sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in det_sd.items()}
ref = orc.ffraft_forward(sd, *inp, raft_iters=2, training=os.environ.get("FREEZE", "1") != "1")
ref[-1].abs().mean().backward()
print("pred err", (preds[-1].detach().cpu() - ref[-1].detach()).abs().max().item(), ref[-1].abs().max().item())
params = dict(m.named_parameters(remove_duplicate=False))
rows = []
for name, p in params.items():
    if p.grad is None or sd.get(name) is None or sd[name].grad is None: continue
    r = sd[name].grad
    e = (p.grad.cpu() - r).abs().max().item(); s = r.abs().max().item()
    if s > 1e-5: rows.append((e / (s + 1e-30), name, e, s))
rows.sort(reverse=True)
for rel, name, e, s in rows[:25]:
    print("%-60s rel %.2e  err %.2e  max %.2e" % (name, rel, e, s))
print("median rel", np.median([r[0] for r in rows]))
