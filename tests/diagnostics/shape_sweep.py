"""One-off robustness sweep: HIP forward vs the CPU oracle over odd frame sizes, batch sizes and fusion types."""
import os, sys, itertools, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
from focusflow_official_amd import FF_RAFT_FUSION
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor

torch.set_num_threads(16)
worst = 0.0
for ft in ("1x1conv", "concat", "SA", "CA"):
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE=ft, LOAD_MODULE_TO_BRANCH=False))
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    sd = {k: det_tensor(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.cuda().eval()
    shapes = [(1, 128, 136), (2, 136, 128), (1, 152, 248), (3, 128, 264), (1, 200, 136), (1, 168, 312)] if ft == "1x1conv" else [(1, 136, 152), (2, 128, 200)]
    for (b, h, w) in shapes:
        inp = orc.shifted_pair(b, h, w, seed=h + w)
        with torch.no_grad():
            fl, fu = m(*[t.cuda() for t in inp], raft_iters=4, test_mode=True)
            rl, ru = orc.ffraft_forward(sd, *inp, raft_iters=4, test_mode=True, fusion_type=ft)
        d = (fu.cpu() - ru).abs().max().item()
        worst = max(worst, d)
        print(f"{ft:8s} B{b} {h}x{w}: max|flow_up - oracle| = {d:.2e}  max|flow| {ru.abs().max().item():.2f}  {'OK' if d < 1e-3 else 'FAIL'}", flush=True)
print("worst", worst)
