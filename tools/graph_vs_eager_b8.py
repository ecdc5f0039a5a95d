"""hipGraph replay against eager launches at the headline shape (8 pairs 384x512, 12 iterations)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from focusflow_official_amd import FF_RAFT_FUSION
from focusflow_official_amd.graph import GraphedForward
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
batch = bench.synthetic_batch(B, 384, 512, 1, dev)
with torch.no_grad():
    for _ in range(3): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); print(f"eager ms/step {(time.perf_counter() - t0) / 20 * 1e3:.3f}")
g = GraphedForward(m, batch, raft_iters=12)
for _ in range(3): g(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g(*batch)
torch.cuda.synchronize(); print(f"graph ms/step {(time.perf_counter() - t0) / 20 * 1e3:.3f}")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); print(f"eager ms/step {(time.perf_counter() - t0) / 20 * 1e3:.3f}")
