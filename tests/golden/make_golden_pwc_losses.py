#!/usr/bin/env python
"""Generate tests/golden/pwc_losses.npz by running the reference's FF-PWC losses (core/models/ff-pwcnet/losses/losses.py,
pure PyTorch: imports unmodified) on random multi-scale predictions.  Stored: loss, 'epe' metric and the gradient
w.r.t. every pyramid level, for EPELoss / CPCL / MixLoss in 'pretrain' (L2) and fine-tune ((L1 + eps)^q) mode.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pwc_losses.py"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/core/models/ff-pwcnet")
sys.dont_write_bytecode = True
from losses import build_losses  # noqa: E402  (reference)

B, H, W = 2, 64, 96
SIZES = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]


def cfg(loss_type, mode, ks, sigma):
    return Namespace(TRAIN=Namespace(LOSS_TYPE=loss_type, LOSS_MODE=mode, LOSS_WEIGHTS=[0.005, 0.01, 0.02, 0.08, 0.32],
                                     LOSS_Q=0.4, LOSS_EPSILON=0.01, LOSS_KERNEL_SIZE=ks, LOSS_SIGMA=sigma, LOSS_LAMDA=0.7))


def main():
    g = torch.Generator().manual_seed(12)
    target = torch.randn(B, 2, H, W, generator=g) * 3
    mask = (torch.rand(B, 1, H, W, generator=g) < 0.03).float() * 255
    preds = [torch.randn(B, 2, h, w, generator=g) for h, w in SIZES]
    out = {"target": target.numpy(), "mask": mask.numpy()}
    for i, p in enumerate(preds):
        out[f"pred{i}"] = p.numpy()
    for lt in ("EPELoss", "CPCL", "MixLoss"):
        for mode in ("pretrain", "finetune"):
            for ks, sigma in ((1, 0.01), (5, 1.7)):
                if lt == "EPELoss" and ks != 1:
                    continue
                crit = build_losses(cfg(lt, mode, ks, sigma))
                ps = [p.clone().requires_grad_(True) for p in preds]
                loss, res = crit(ps, target, False) if lt == "EPELoss" else crit(ps, target, mask, False)
                loss.backward()
                tag = f"{lt}_{mode}_k{ks}"
                out[tag + "_loss"] = np.array([loss.item(), float(res["epe"])], dtype=np.float64)
                for i, p in enumerate(ps):
                    out[f"{tag}_grad{i}"] = p.grad.numpy()
                print(tag, loss.item(), float(res["epe"]))
    # sparse ground truth (KITTI stage: train.py:287-312 calls the loss with sparse=True): EPELoss and MixLoss (CPCL raises
    # in the reference itself).  About a third of the target pixels are "invalid" = exactly (0, 0).
    drop = torch.rand(B, 1, H, W, generator=g) < 0.35
    starget = torch.where(drop, torch.zeros_like(target), target)
    out["sparse_target"] = starget.numpy()
    for lt in ("EPELoss", "MixLoss"):
        for mode in ("pretrain", "finetune"):
            ks, sigma = (1, 0.01) if lt == "EPELoss" else (5, 1.7)
            crit = build_losses(cfg(lt, mode, ks, sigma))
            ps = [p.clone().requires_grad_(True) for p in preds]
            loss, res = crit(ps, starget, True) if lt == "EPELoss" else crit(ps, starget, mask, True)
            loss.backward()
            tag = f"sparse_{lt}_{mode}_k{ks}"
            out[tag + "_loss"] = np.array([loss.item(), float(res["epe"])], dtype=np.float64)
            for i, p in enumerate(ps):
                out[f"{tag}_grad{i}"] = p.grad.numpy()
            print(tag, loss.item(), float(res["epe"]))
    try:
        build_losses(cfg("CPCL", "pretrain", 5, 1.7))([p.clone() for p in preds], starget, mask, True)
        out["sparse_CPCL_raises"] = np.array([0])
    except Exception as e:  # noqa: BLE001
        print("CPCL sparse in the reference raises:", type(e).__name__, str(e)[:80])
        out["sparse_CPCL_raises"] = np.array([1])
    np.savez_compressed(os.path.join(HERE, "pwc_losses.npz"), **out)


if __name__ == "__main__":
    main()
