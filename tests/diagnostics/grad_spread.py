"""Diagnostic: for every parameter, |hip - fp64| and |fp32 oracle - fp64| (relative to the tensor's max) on one
frozen-BN training step.  FF_CONV_PRECISION=fp32 switches the HIP convolutions to exact fp32 MFMA."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from focusflow_official_amd import FF_RAFT_FUSION
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor
import test_hip_backward as T

H, W, IT = int(os.environ.get("H", 128)), int(os.environ.get("W", 128)), int(os.environ.get("IT", 2))
spec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "state_dict_spec.json")))
sd = {k: det_tensor(k, s) for k, s, _ in spec}
NOISE = os.environ.get("MODE") == "noiseframe"
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="frame" if NOISE else "point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
m.load_state_dict(sd, strict=True)
m = m.cuda().train()
m.flow_net.freeze_bn()
inp = orc.shifted_pair(1, H, W, seed=9)
if NOISE:
    gg = torch.Generator().manual_seed(1)
    inp = (torch.rand(1, 3, H, W, generator=gg) * 255, torch.rand(1, 3, H, W, generator=gg) * 255, inp[2], inp[3])
loss_fn = lambda ref: sum(p.abs().mean() for p in ref)
preds = m(*[t.cuda() for t in inp], raft_iters=IT)
loss_fn(preds).backward()
torch.set_num_threads(16)
if NOISE:
    def og(dtype):
        s2 = {k: ((v.to(dtype).clone().requires_grad_(True) if "running_" not in k else v.to(dtype).clone()) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        a, b, _, _ = orc.prepare_inputs(*inp, 3)
        ref = orc.raft_forward(s2, a.to(dtype), b.to(dtype), a.to(dtype), b.to(dtype), IT, None, False, False, "1x1conv", prefix="flow_net.")
        loss_fn(ref).backward()
        return {k: v.grad for k, v in s2.items() if getattr(v, "grad", None) is not None}, ref[-1].detach()
    g32, last32 = og(torch.float32)
    g64, last64 = og(torch.float64)
else:
    g32, last32 = T._oracle_grads(sd, inp, IT, loss_fn, torch.float32)
    g64, last64 = T._oracle_grads(sd, inp, IT, loss_fn, torch.float64)
print("forward: |hip - fp64|", float((preds[-1].detach().cpu().double() - last64).abs().max()), "|fp32 - fp64|", float((last32.double() - last64).abs().max()))
rows = []
for k, p in m.named_parameters():
    if p.grad is None or k not in g64:
        continue
    w = g64[k]; s = float(w.abs().max())
    if s < 1e-7:
        continue
    rows.append((float((p.grad.cpu().double() - w).abs().max()) / s, float((g32[k].double() - w).abs().max()) / s, k))
order = [k for k, _ in m.named_parameters()]
for hip, ref, k in rows:
    flag = " <<<" if hip > 3 * ref + 1e-4 else ""
    print(f"{hip:9.2e} {ref:9.2e}  {k}{flag}")
if os.environ.get("DUMP"):
    keys = os.environ["DUMP"].split(",")
    params = dict(m.named_parameters())
    torch.save({k: (params[k].grad.cpu(), g32[k], g64[k]) for k in keys}, os.environ.get("DUMP_TO", "gpurun_out/grad_dump.pt"))
