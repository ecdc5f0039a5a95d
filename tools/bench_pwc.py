"""FF-PWC forward timing at BASELINE config 4 (B=1, 448x1024, SIFT-like mask)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from focusflow_official_amd.pwcnet import FF_PWCNET
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
torch.manual_seed(0)
m = FF_PWCNET(cfg).cuda().eval()
with torch.no_grad():   # keep activations in fp16 range for unnormalised inputs
    m.netExtractor.netOne[0].weight.mul_(1 / 255.0); m.netExtractor.mask_netOne[0].weight.mul_(1 / 255.0)
b = int(os.environ.get("B", 1))
g = torch.Generator().manual_seed(0)
i1 = torch.randint(0, 256, (b, 3, 448, 1024), generator=g).float().cuda()
i2 = torch.roll(i1, (3, -5), (2, 3))
m1 = ((torch.rand(b, 1, 448, 1024, generator=g) < 2000 / (448 * 1024)).float() * 255).cuda()
with torch.no_grad():
    for _ in range(3):
        out = m(i1, i2, m1, m1, test_mode=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = m(i1, i2, m1, m1, test_mode=True)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"FF-PWC forward B={b} 448x1024: {dt * 1e3:.2f} ms/step = {b / dt:.1f} frame-pairs/s, finite={bool(torch.isfinite(out).all())}", file=sys.stderr)
import json
print(json.dumps({"metric": "frame-pairs/sec FF-PWC forward 448x1024 (BASELINE configs[3])", "value": round(b / dt, 2), "unit": "frame-pairs/s",
                  "n_gpus": 1, "steps": n, "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "f32 via fp16x3 split operands",
                  "data": "synthetic", "config": {"workload": f"FF_PWCNET forward (test_mode), {b} pair(s) 448x1024, SIFT-like mask (2000 points), random-init weights"}}))
