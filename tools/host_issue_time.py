"""How long does the host need to ISSUE one forward step (no sync), vs. the GPU time of that step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from focusflow_official_amd import FF_RAFT_FUSION
dev = torch.device("cuda", 0)
model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
for b in (1, 8):
    batch = bench.synthetic_batch(b, 384, 512, 1, dev)
    with torch.no_grad():
        for _ in range(3):
            model(*batch, raft_iters=12, test_mode=True)
        torch.cuda.synchronize()
        issue, total = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            model(*batch, raft_iters=12, test_mode=True)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append(t1 - t0); total.append(t2 - t0)
    print(f"B={b}: host issue {min(issue)*1e3:.2f} ms, issue+drain {min(total)*1e3:.2f} ms")
