"""Which Python lines of the forward issue device-to-device copies (aten::copy_) - each is a 4-5 us launch in the update loop."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from torch.profiler import profile, ProfilerActivity
from focusflow_official_amd import FF_RAFT_FUSION

cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg).cuda().eval()
g = torch.Generator().manual_seed(0)
im = [torch.randint(0, 256, (8, 3, 384, 512), generator=g).float().cuda() for _ in range(2)]
mk = [((torch.rand(8, 1, 384, 512, generator=g) < 0.0025).float() * 255).cuda() for _ in range(2)]
with torch.no_grad():
    for _ in range(2):
        m(im[0], im[1], mk[0], mk[1], raft_iters=12, test_mode=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        m(im[0], im[1], mk[0], mk[1], raft_iters=12, test_mode=True)
        torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::clone", "aten::contiguous"):
        st = [s for s in e.stack if "focusflow_official_amd" in s or "bench" in s][:2]
        cnt[(e.name, " <- ".join(s.split("/")[-1] for s in st))] += 1
for (n, s), c in cnt.most_common(30):
    print(f"{c:4d}  {n:18s} {s}")
