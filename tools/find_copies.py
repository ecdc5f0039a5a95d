"""Which Python lines issue device fills / copies / adds (each a 4-7 us launch): MODE=fwd (inference forward, default) or
MODE=train (one training step: forward, MixLoss, backward)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from torch.profiler import profile, ProfilerActivity
from focusflow_official_amd import FF_RAFT_FUSION

train = os.environ.get("MODE", "fwd") == "train"
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg).cuda()
m = m.train() if train else m.eval()
h, w = (368, 496) if train else (384, 512)
g = torch.Generator().manual_seed(0)
im = [torch.randint(0, 256, (8, 3, h, w), generator=g).float().cuda() for _ in range(2)]
mk = [((torch.rand(8, 1, h, w, generator=g) < 0.0025).float() * 255).cuda() for _ in range(2)]
if train:
    from focusflow_official_amd.losses import build_losses
    crit = build_losses("MixLoss", gamma=0.8, max_flow=400, kernel_size=1, sigma=0.01, lamda=1)
    gt = torch.randn(8, 2, h, w, generator=g).cuda() * 5
    valid = torch.ones(8, h, w).cuda()

    def step():
        preds = m(im[0], im[1], mk[0], mk[1], raft_iters=12)
        loss, _ = crit(preds, gt, valid, mk[0])
        m.zero_grad(set_to_none=True)
        loss.backward()
else:
    def step():
        with torch.no_grad():
            m(im[0], im[1], mk[0], mk[1], raft_iters=12, test_mode=True)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
names = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::zeros_like", "aten::mul")
for e in prof.events():
    if e.name in names:
        st = [s for s in e.stack if "focusflow_official_amd" in s or "bench" in s or "find_copies" in s][:2]
        cnt[(e.name, " <- ".join(s.split("/")[-1] for s in st))] += 1
for (n, s), c in cnt.most_common(40):
    print(f"{c:4d}  {n:18s} {s}")
