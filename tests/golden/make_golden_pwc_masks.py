#!/usr/bin/env python
"""FF-PWC mask modes other than 'point' (authoring container only): the reference's own `init_mask`
(core/models/ff-pwcnet/PWCNet_Core/ff_pwcnet.py:61-110) and the whole FF_PWCNET forward in 'frame' and 'neighborG'
mode (the two that never call cv2), with the same stand-ins as make_golden_pwc.py (empty `cv2`; `correlation` =
the oracle's cost_volume; `.cuda()` a no-op).  'neighborE' / 'context' need cv.getStructuringElement and stay pinned
by definition only (the ellipse table restates OpenCV's algorithm).
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pwc_masks.py"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_pwc as base  # noqa: E402  (installs the stand-ins, imports the reference's FF_PWCNET)
from PWCNet_Core.ff_pwcnet import init_mask  # noqa: E402  (reference)


def main():
    out = {}
    for modal in ("frame", "neighborG"):
        cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL=modal, KERNEL_SIZE=9, KERNEL_SIGMA=1.5, MASK_DILATE=7),
                        MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
        net = base.FF_PWCNET(cfg)
        net.load_state_dict(base.weights(net), strict=True)
        net.eval()
        for tag, (b, h, w) in (("128x192", (1, 128, 192)), ("100x180", (1, 100, 180))):
            i1, i2, m1 = base.inputs(b, h, w, seed=14 if tag == "128x192" else 15)
            with torch.no_grad():
                p1, p2, pm1, pm2 = net.preprocess(i1, i2, m1, torch.zeros_like(m1))
                im1, im2 = init_mask(p1, p2, pm1, pm2, cfg)
                flows = net(i1, i2, m1, torch.zeros_like(m1))
                full = net(i1, i2, m1, torch.zeros_like(m1), test_mode=True)
            out[f"{modal}_{tag}_in_crc"] = np.array([base.crc(i1), base.crc(i2), base.crc(m1)], dtype=np.int64)
            out[f"{modal}_{tag}_mask1"] = im1[:, :1].numpy().astype(np.float32) if modal != "frame" else np.zeros(1, np.float32)
            out[f"{modal}_{tag}_mask2_minmax"] = np.array([im2.min().item(), im2.max().item()])
            out[f"{modal}_{tag}_flow2"] = flows[0].numpy().astype(np.float32)
            out[f"{modal}_{tag}_full"] = full.numpy().astype(np.float32)
            print(modal, tag, "max|flow|", float(full.abs().max()), "mask1 max", float(im1.max()))
    np.savez_compressed(os.path.join(HERE, "pwc_mask_modes.npz"), **out)


if __name__ == "__main__":
    main()
