import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from focusflow_official_amd import FF_RAFT_FUSION
from focusflow_official_amd.graph import GraphedForward
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
batch = bench.synthetic_batch(1, 384, 512, 1, dev)
with torch.no_grad():
    for _ in range(3): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); print("eager ms/pair", (time.perf_counter() - t0) / 20 * 1e3)
g = GraphedForward(m, batch, raft_iters=12)
for _ in range(3): g(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g(*batch)
torch.cuda.synchronize(); print("graph ms/pair", (time.perf_counter() - t0) / 50 * 1e3)
