"""cProfile of the HOST side of one forward (launches only, no sync inside): where the 7 ms of a one-pair step go."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.getcwd())
import torch
import bench
from focusflow_official_amd import FF_RAFT_FUSION
dev = torch.device("cuda:0")
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = bench.synthetic_batch(B, 384, 512, 1, dev)
with torch.no_grad():
    for _ in range(3): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): m(*batch, raft_iters=12, test_mode=True)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
