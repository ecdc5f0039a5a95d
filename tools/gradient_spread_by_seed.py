import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from argparse import Namespace
from focusflow_official_amd import FF_RAFT_FUSION
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
det_sd = {k: det_tensor(k, s) for k, s, _ in json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_spec.json")))}
def oracle_grads(inp, iters, loss_fn, dtype):
    sd = {k: ((v.to(dtype).clone().requires_grad_(True) if "running_" not in k else v.to(dtype).clone()) if v.is_floating_point() else v.clone()) for k, v in det_sd.items()}
    ref = orc.ffraft_forward(sd, *[t.to(dtype) for t in inp], raft_iters=iters, training=False)
    loss_fn(ref).backward()
    return {k: v.grad for k, v in sd.items() if getattr(v, "grad", None) is not None}
for seed in (9, 10, 11, 12):
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    m.load_state_dict(det_sd, strict=True)
    m = m.cuda().train()
    m.flow_net.freeze_bn()
    inp = orc.shifted_pair(1, 128, 128, seed=seed)
    preds = m(*[t.cuda() for t in inp], raft_iters=2)
    preds[-1].abs().mean().backward()
    loss_fn = lambda ref: ref[-1].abs().mean()
    g32 = oracle_grads(inp, 2, loss_fn, torch.float32)
    g64 = oracle_grads(inp, 2, loss_fn, torch.float64)
    hip, cpu = [], []
    for k, p in m.named_parameters(remove_duplicate=False):
        if p.grad is None or k not in g64 or float(g64[k].abs().max()) < 1e-7: continue
        s = float(g64[k].abs().max())
        hip.append(float((p.grad.cpu().double() - g64[k]).abs().max()) / s)
        cpu.append(float((g32[k].double() - g64[k]).abs().max()) / s)
    hip, cpu = np.array(hip), np.array(cpu)
    print(f"seed {seed}: HIP median {np.median(hip):.2e} p90 {np.percentile(hip, 90):.2e} max {hip.max():.2e} | CPU fp32 median {np.median(cpu):.2e} p90 {np.percentile(cpu, 90):.2e} max {cpu.max():.2e}", flush=True)
