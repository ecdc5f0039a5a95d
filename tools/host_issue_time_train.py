"""How long does the host need to ISSUE one training step (no sync), vs. the step's wall time?  Split by phase."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from focusflow_official_amd import FF_RAFT_FUSION
from focusflow_official_amd.losses import build_losses
dev = torch.device("cuda", 0)
B, h, w = 8, 368, 496
torch.manual_seed(1234)
model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=4e-4, weight_decay=1e-5, eps=1e-8)
crit = build_losses("MixLoss", gamma=0.8, max_flow=400, kernel_size=1, sigma=0.01, lamda=1)
batch = bench.synthetic_batch(B, h, w, 1234, dev)
flow_gt = (torch.randn(B, 2, h, w) * 5).clamp(-400, 400).to(dev)
valid = torch.ones(B, h, w, device=dev)
sync = torch.cuda.synchronize


def step(with_sync):
    ts = [time.perf_counter()]
    def mark():
        if with_sync:
            sync()
        ts.append(time.perf_counter())
    preds = model(*batch, raft_iters=12); mark()
    loss, _ = crit(preds, flow_gt, valid, batch[2]); mark()
    opt.zero_grad(set_to_none=True)
    loss.backward(); mark()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); mark()
    opt.step(); mark()
    return [b - a for a, b in zip(ts, ts[1:])]


for _ in range(4):
    step(False)
sync()
names = ["forward", "loss", "backward", "clip", "adamw"]
for with_sync in (False, True):
    best = None
    for _ in range(5):
        sync()
        t0 = time.perf_counter()
        d = step(with_sync)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, d)
    print(("phases drained one by one (GPU time of each)" if with_sync else "host issue only (no sync inside)") +
          f": step {best[0]*1e3:.1f} ms, host returned after {best[1]*1e3:.1f} ms; " +
          ", ".join(f"{n} {x*1e3:.1f}" for n, x in zip(names, best[2])))
