"""Where a step's time is, without a profiler: HIP events around the encoder part and the update loop of the 8-pair forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from focusflow_official_amd import FF_RAFT_FUSION, raft_net

dev = torch.device("cuda", 0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
batch = bench.synthetic_batch(b, 384, 512, 1, dev)
ev = []
orig = raft_net.RAFT._loop


def timed_loop(self, *a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig(self, *a, **k)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    ev.append((e0, e1))
    return r


raft_net.RAFT._loop = timed_loop
enc = {}          # name -> [(start, end)] of the two encoders, each on the stream it runs on
from focusflow_official_amd import cce
orig_fwd = cce.BasicParallelFusionLayer.forward


def timed_enc(self, x, mask):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig_fwd(self, x, mask)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    enc.setdefault(self.norm_fn, []).append((e0, e1))
    return r


cce.BasicParallelFusionLayer.forward = timed_enc
steps = []
with torch.no_grad():
    for _ in range(3):
        model(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize()
    ev.clear()
    enc.clear()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    n = 10
    for _ in range(n):
        t = torch.cuda.Event(enable_timing=True); t.record()
        steps.append(t)
        model(*batch, raft_iters=12, test_mode=True)
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    torch.cuda.synchronize()
loop = sum(a.elapsed_time(b_) for a, b_ in ev) / n
total = s0.elapsed_time(s1) / n
print(f"B={b}: step {total:.3f} ms = encoders + corr build {total - loop:.3f} ms + update loop {loop:.3f} ms ({loop / 12 * 1e3:.1f} us per iteration)")
for name, what in (("batch", "context encoder (side stream)"), ("instance", "feature encoder (main stream)")):
    st = sum(t.elapsed_time(a) for t, (a, _) in zip(steps, enc[name])) / n
    en = sum(t.elapsed_time(b_) for t, (_, b_) in zip(steps, enc[name])) / n
    print(f"   {what}: from {st:.3f} to {en:.3f} ms after the step's first launch")
lp = sum(t.elapsed_time(a) for t, (a, _) in zip(steps, ev)) / n
print(f"   update loop starts at {lp:.3f} ms")
