#!/usr/bin/env python
"""Golden vectors for the loss module, produced by RUNNING the reference's
core/models/ff-raft/losses/losses.py (pure torch/numpy: importable here).  Authoring container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/core/models/ff-raft")
sys.dont_write_bytecode = True
from losses import build_losses  # noqa: E402  (reference)
from oracle import ffraft_ref as orc  # noqa: E402

rec = {}
for kind, kw in (("EPELoss", {}), ("CPCL", dict(kernel_size=5, sigma=1.7)), ("MixLoss", dict(kernel_size=5, sigma=1.7, lamda=0.8)),
                 ("MixLoss_k1", dict(kernel_size=1, sigma=0.01, lamda=1))):       # last: ffraft_chairs_orb.yaml:35-39
    preds, gt, valid, mask = orc.loss_inputs()
    preds = [p.requires_grad_(True) for p in preds]
    fn = build_losses(kind.split("_")[0], gamma=0.8, max_flow=400, **kw)
    loss, metrics = fn(preds, gt, valid, mask)
    loss.backward()
    rec[kind + ":loss"] = np.array([loss.item()], np.float64)
    rec[kind + ":epe"] = np.array([metrics["epe"]], np.float64)
    for i, p in enumerate(preds):
        rec[f"{kind}:g{i}"] = p.grad[:, :, ::3, ::3].numpy().copy()
np.savez_compressed(os.path.join(HERE, "losses.npz"), **rec)
print({k: v.shape for k, v in rec.items() if "loss" in k or "epe" in k}, [float(rec[k][0]) for k in rec if k.endswith(":loss")])
