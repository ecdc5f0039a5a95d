"""Where a training step's GPU time is, phase by phase (each phase drained: synchronize before and after):
encoders + corr build | update loop forward | loss | update loop backward | rest of the backward (encoders) | clip + AdamW."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from focusflow_official_amd import FF_RAFT_FUSION, train_loop
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
T = {}
F = train_loop.UpdateLoopFn
f0, b0 = F.forward, F.backward


def timed(name, f):
    def g(*a, **k):
        sync(); t = time.perf_counter(); r = f(*a, **k); sync(); T[name] = T.get(name, 0.0) + time.perf_counter() - t
        return r
    return g


F.forward = staticmethod(timed("loop_fwd", f0))
F.backward = staticmethod(timed("loop_bwd", b0))


def step():
    sync(); t0 = time.perf_counter()
    preds = model(*batch, raft_iters=12); sync(); t1 = time.perf_counter()
    loss, _ = crit(preds, flow_gt, valid, batch[2]); sync(); t2 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss.backward(); sync(); t3 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step(); sync(); t4 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t4 - t3


for _ in range(4):
    step()
n = 5
acc = [0.0] * 4
T.clear()
for _ in range(n):
    for i, v in enumerate(step()):
        acc[i] += v
fw, ls, bw, op = [a / n * 1e3 for a in acc]
lf, lb = T["loop_fwd"] / n * 1e3, T["loop_bwd"] / n * 1e3
print(f"forward {fw:.1f} ms = encoders + corr build {fw - lf:.1f} + update loop {lf:.1f}; loss {ls:.1f}; "
      f"backward {bw:.1f} = update loop {lb:.1f} + encoders etc. {bw - lb:.1f}; clip + AdamW {op:.1f}; sum {fw + ls + bw + op:.1f} ms")
