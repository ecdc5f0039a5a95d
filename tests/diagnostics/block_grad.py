"""Diagnostic: one residual block (fnet.layer2.1: 96 ch, InstanceNorm) forward + backward on the HIP path vs the oracle in
fp64 and fp32, for a smooth (structured) input / upstream gradient and for white noise."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
import torch.nn.functional as F
from focusflow_official_amd import FF_RAFT_FUSION
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor

spec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "state_dict_spec.json")))
sd = {k: det_tensor(k, s) for k, s, _ in spec}
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
m.load_state_dict(sd, strict=True)
m = m.cuda().train()
enc = m.flow_net.fnet
for blkname, c, hw in (("layer2.1", 96, 32), ("layer1.0", 64, 64)):
    blk = dict(enc.named_modules())[blkname]
    pre = "flow_net.fnet." + blkname
    for kind in ("smooth", "noise", "tiny", "outliers", "tiny+outliers"):
        g = torch.Generator().manual_seed(7)
        if kind == "smooth":
            x = F.interpolate(torch.randn(1, c, hw // 4, hw // 4, generator=g), size=(hw, hw), mode="bilinear").relu() + 0.01 * torch.randn(1, c, hw, hw, generator=g).abs()
            G = F.interpolate(torch.randn(1, c, hw // 8, hw // 8, generator=g), size=(hw, hw), mode="bilinear")
        else:
            x = torch.randn(1, c, hw, hw, generator=g).relu()
            G = torch.randn(1, c, hw, hw, generator=g)
            if "tiny" in kind:
                G = G * 1e-7
            if "outliers" in kind:
                idx = torch.randint(0, G.numel(), (20,), generator=g)
                G.view(-1)[idx] *= 3e4
        res = {}
        for dt in (torch.float64, torch.float32):
            s2 = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items() if k.startswith(pre) and v.is_floating_point()}
            xx = x.to(dt).clone().requires_grad_(True)
            y = orc._resblock(s2, pre, xx, "instance", 1, False)
            (y * G.to(dt)).sum().backward()
            res[dt] = (y.detach(), xx.grad, s2[pre + ".conv1.weight"].grad, s2[pre + ".conv2.weight"].grad)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        yd = enc._block(blk, xd)
        (yd * G.permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
        hip = (yd.detach().permute(0, 3, 1, 2).cpu(), xd.grad.permute(0, 3, 1, 2).cpu(), blk.conv1.weight.grad.cpu(), blk.conv2.weight.grad.cpu())
        for name, h, r32, r64 in zip(("y", "dx", "dW1", "dW2"), hip, res[torch.float32], res[torch.float64]):
            s = float(r64.abs().max())
            print(f"{blkname} {kind:6s} {name:4s}: |hip-fp64|/max {float((h.double() - r64).abs().max()) / s:.2e}   |fp32-fp64|/max {float((r32.double() - r64).abs().max()) / s:.2e}")
