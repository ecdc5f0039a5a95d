#!/usr/bin/env python
"""Generate tests/golden/wrapper_*.npz by RUNNING THE REFERENCE'S FF_RAFT_FUSION wrapper (authoring container only).

`FF_RAFT_Core/ff_raft.py` imports cv2 at module scope; this image has no OpenCV, so an EMPTY stand-in module is put
in sys.modules.  The mask modes that never touch cv2 — 'point', 'frame', 'neighborG' (ff_raft.py:31-38, 53-70) — then
run unmodified: `init_mask`, the [0,255] -> [-1,1] scaling (:142-145) and the whole forward.  'neighborE' / 'context'
call cv.getStructuringElement (:26,45) and cannot be produced here.

Stored: the two prepared mask tensors of init_mask, and flow_low / flow_up of FF_RAFT_FUSION.forward in test_mode.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_wrapper.py
"""
import os
import sys
import types
import zlib
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/core/models/ff-raft")
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))     # absent here; the modes below never call it

from FF_RAFT_Core.ff_raft import FF_RAFT_FUSION, init_mask  # noqa: E402  (reference)

from oracle import ffraft_ref as orc  # noqa: E402
from oracle.weights import det_tensor  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def crc(t):
    return zlib.crc32(t.contiguous().numpy().tobytes())


def main():
    out = {}
    for modal in ("point", "frame", "neighborG"):
        cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL=modal, MASK_DILATE=31, KERNEL_SIZE=31, KERNEL_SIGMA=5),
                        MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
        net = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
        net.load_state_dict({k: det_tensor(k, v.shape) for k, v in net.state_dict().items()}, strict=True)
        net.eval()
        inp = orc.shifted_pair(2, 128, 160, seed=17)
        with torch.no_grad():
            m1, m2 = init_mask(inp[0].contiguous(), inp[1].contiguous(), inp[2], inp[3], cfg)
            fl, fu = net(*inp, raft_iters=3, test_mode=True)
        out[f"{modal}_mask1"] = m1.float().numpy().astype(np.float32)[:, :, ::2, ::2]     # subsampled: 2x2 grid
        out[f"{modal}_mask2"] = m2.float().numpy().astype(np.float32)[:, :, ::2, ::2]
        out[f"{modal}_flow_low"] = fl.numpy().astype(np.float32)
        out[f"{modal}_flow_up"] = fu.numpy().astype(np.float32)
        print(modal, "mask1 range", float(m1.min()), float(m1.max()), "max|flow|", float(fu.abs().max()))
    out["in_crc"] = np.array([crc(t) for t in inp], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "wrapper_modes_128x160_b2_it3.npz"), **out)


if __name__ == "__main__":
    main()
