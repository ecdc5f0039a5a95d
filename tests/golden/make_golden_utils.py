#!/usr/bin/env python
"""Generate tests/golden/utils_padder.npz with the reference's InputPadder / forward_interpolate (core/utils/utils.py).
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_utils.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/core")
sys.dont_write_bytecode = True
from utils.utils import InputPadder, forward_interpolate  # noqa: E402  (reference)

out = {}
g = torch.Generator().manual_seed(3)
for i, (h, w, mode) in enumerate([(436, 1024, "sintel"), (375, 1242, "kitti"), (370, 1226, "kitti"), (100, 180, "sintel"), (128, 192, "sintel")]):
    x = torch.randn(1, 2, h, w, generator=g)
    p = InputPadder(x.shape, mode=mode)
    y = p.pad(x)[0]
    out[f"case{i}"] = np.array([h, w, y.shape[-2], y.shape[-1]] + list(p._pad), dtype=np.int64)
    out[f"case{i}_corner"] = y[0, :, :12, :12].numpy()
    out[f"case{i}_unpad_ok"] = np.array([int(torch.equal(p.unpad(y), x))])
    out[f"case{i}_x_corner"] = x[0, :, :12, :12].numpy()
    out[f"case{i}_seedcheck"] = np.array([float(x.sum())])
f = torch.randn(2, 14, 18, generator=g) * 2
out["fi_in"] = f.numpy()
out["fi_out"] = forward_interpolate(f).numpy()
np.savez_compressed(os.path.join(HERE, "utils_padder.npz"), **out)
print({k: v.tolist() for k, v in out.items() if k.startswith("case") and k[-1].isdigit()})
