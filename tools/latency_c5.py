"""BASELINE configs[4] at one pair (544x960, 32 iterations, fp16 pyramid): eager launches against hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from focusflow_official_amd import FF_RAFT_FUSION
from focusflow_official_amd.graph import GraphedForward
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
m.flow_net.corr_pyramid_dtype = "fp16"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = bench.synthetic_batch(B, 544, 960, 1, dev)
with torch.no_grad():
    for _ in range(3): m(*batch, raft_iters=32, test_mode=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(*batch, raft_iters=32, test_mode=True)
    torch.cuda.synchronize(); print(f"eager ms/step {(time.perf_counter() - t0) / 10 * 1e3:.3f}")
    t0 = time.perf_counter()
    for _ in range(10): m(*batch, raft_iters=32, test_mode=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); print(f"host issue ms/step {(t1 - t0) / 10 * 1e3:.3f}")
g = GraphedForward(m, batch, raft_iters=32)
for _ in range(3): g(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g(*batch)
torch.cuda.synchronize(); print(f"graph ms/step {(time.perf_counter() - t0) / 10 * 1e3:.3f}")
