"""FF-PWC forward, 8 pairs 448x1024 (BASELINE configs[3]) a few times - the command rocprofv3 --kernel-trace --stats wraps.

    python tools/pwc_b8.py [steps]
"""
import os
import sys
import time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focusflow_official_amd.pwcnet import FF_PWCNET  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
pcfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
torch.manual_seed(0)
m = FF_PWCNET(pcfg).to(dev).eval()
with torch.no_grad():       # (unnormalised 0..255 inputs, as bench.py's leg: keep the first layer's activations in fp16 range)
    m.netExtractor.netOne[0].weight.mul_(1 / 255.0)
    m.netExtractor.mask_netOne[0].weight.mul_(1 / 255.0)
g = torch.Generator().manual_seed(0)
b = 8
im1 = torch.randint(0, 256, (b, 3, 448, 1024), generator=g).float().to(dev)
im2 = torch.roll(im1, (3, -5), (2, 3))
mask = ((torch.rand(b, 1, 448, 1024, generator=g) < 2000 / (448 * 1024)).float() * 255).to(dev)
with torch.no_grad():
    for _ in range(3):
        m(im1, im2, mask, mask, test_mode=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m(im1, im2, mask, mask, test_mode=True)
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step of 8 pairs")
