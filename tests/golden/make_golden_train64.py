#!/usr/bin/env python
"""fp64 companion of the train-step fixture (train_shift_128x128_b2_it3.npz, written by make_golden.py from the
reference): the same training step - train-mode forward (BatchNorm batch statistics), sequence L1, backward -
evaluated in DOUBLE by the oracle, for the same 14 sampled parameter gradients.

The reference itself cannot run in double (raft.py:191-193 and corr.py:50 cast to fp32 internally); the oracle keeps
fp64 for exactly this and is first held here to the reference's fp32 gradients of the committed fixture.  The GPU test
then bounds |hip - fp64| by the reference's own |fp32 - fp64| instead of a blanket 1 % (tests/test_hip_backward.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train64.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ffraft_ref as orc  # noqa: E402
from oracle.weights import det_tensor  # noqa: E402

torch.set_num_threads(8)


def step(sd, inp, flow_gt, valid, dtype):
    sd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone().to(dtype) if v.is_floating_point() else v.clone())
          for k, v in sd.items()}
    preds = orc.ffraft_forward(sd, *[t.to(dtype) for t in inp], raft_iters=3, training=True)
    loss, _ = orc.sequence_l1(preds, flow_gt.to(dtype), valid.to(dtype))
    loss.backward()
    return loss, preds, sd


def main():
    with open(os.path.join(HERE, "state_dict_spec.json")) as f:
        spec = json.load(f)
    sd = {k: det_tensor(k, s) for k, s, _ in spec}
    g = dict(np.load(os.path.join(HERE, "train_shift_128x128_b2_it3.npz")))
    inp = orc.shifted_pair(2, 128, 128, seed=4)
    gen = torch.Generator().manual_seed(5)
    flow_gt = (torch.randn(2, 2, 128, 128, generator=gen) * 5).clamp(-400, 400)
    valid = torch.ones(2, 128, 128)
    l32, p32, sd32 = step(sd, inp, flow_gt, valid, torch.float32)
    l64, p64, sd64 = step(sd, inp, flow_gt, valid, torch.float64)
    rec = {"loss64": np.array([l64.item()])}
    for key in [k for k in g if k.startswith("grad:")]:
        name = "flow_net." + key[5:]
        if sd32[name].grad is None:      # norm3 and downsample.1 are ONE module in the reference (two state_dict keys)
            name = name.replace(".norm3.", ".downsample.1.")
        g32, g64 = sd32[name].grad, sd64[name].grad
        s32 = g32.flatten()[:: max(1, g32.numel() // 512)].numpy()
        s64 = g64.flatten()[:: max(1, g64.numel() // 512)].numpy()
        # the oracle's fp32 step must BE the reference's (same ATen kernels): the committed fixture pins it
        np.testing.assert_allclose(s32, g[key], rtol=0, atol=2e-5 * float(np.abs(g[key]).max()), err_msg=name)
        rec["grad64:" + key[5:]] = s64
        rec["gnorm64:" + key[5:]] = np.array([g64.norm().item()])
        spread = np.abs(g[key].astype(np.float64) - s64).max() / np.abs(s64).max()
        print(f"{name:60s} |fp32ref - fp64| / max = {spread:.2e}")
    np.savez_compressed(os.path.join(HERE, "train_shift_128x128_b2_it3_fp64.npz"), **rec)
    print("loss fp32", l32.item(), "fp64", l64.item())


if __name__ == "__main__":
    main()
