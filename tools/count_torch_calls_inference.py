import collections, os, sys, traceback
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import torch, bench
counts = collections.Counter(); active = False
def caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "count_inf" in fr.filename or "/torch/" in fr.filename: continue
        return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
    return "?"
def wrap(obj, name, label):
    orig = getattr(obj, name)
    def f(*a, **k):
        if active: counts[(label, caller())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("zeros", "zeros_like", "cat", "stack", "empty_like", "ones", "full"): wrap(torch, n, n)
for n in ("zero_", "fill_", "copy_", "contiguous", "clone", "add_", "float", "__add__", "__mul__", "__iadd__", "to", "permute"): wrap(torch.Tensor, n, "Tensor." + n)
from focusflow_official_amd import FF_RAFT_FUSION
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg()).to(dev).eval()
batch = bench.synthetic_batch(8, 384, 512, 1, dev)
with torch.no_grad():
    for _ in range(2): m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); active = True
    m(*batch, raft_iters=12, test_mode=True)
    torch.cuda.synchronize(); active = False
for (lab, where), c in counts.most_common(30): print(f"{c:5d}  {lab:22s} {where}")
