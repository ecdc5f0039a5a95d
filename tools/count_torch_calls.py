"""Which Python lines issue the small torch kernels of one FF-RAFT training step (fills, copies, adds, cats)?
Wraps the torch entry points that launch them and counts calls by caller (file:line), over one step after warm-up.
   python tools/count_torch_calls.py            (on the GPU box)"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

counts = collections.Counter()
active = False


def caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "count_torch_calls" in fr.filename or "/torch/" in fr.filename:
            continue
        return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
    return "?"


def wrap(obj, name, label):
    orig = getattr(obj, name)

    def f(*a, **k):
        if active:
            counts[(label, caller())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for n in ("zeros", "zeros_like", "cat", "stack", "empty_like", "ones", "full"):
    wrap(torch, n, n)
for n in ("zero_", "fill_", "copy_", "contiguous", "clone", "add_", "float", "double", "__add__", "__mul__", "__iadd__"):
    wrap(torch.Tensor, n, "Tensor." + n)

import argparse  # noqa: E402
args = argparse.Namespace(height=384, width=512, batch=8, iters=12, warmup=2, steps=1)
device = torch.device("cuda:0")
torch.cuda.set_device(device)
step, h, w = bench.train_setup(args, 1, 0, 0, device)
for _ in range(2):
    step()
torch.cuda.synchronize()
active = True
step()
torch.cuda.synchronize()
active = False
tot = collections.Counter()
for (lab, where), c in counts.items():
    tot[lab] += c
print("totals:", dict(tot))
for (lab, where), c in counts.most_common(60):
    print(f"{c:5d}  {lab:22s} {where}")
