#!/usr/bin/env python
"""BASELINE configs[4] fixture (authoring container only): the REFERENCE's RAFT.forward (raft.py:173-236) at 544x960
(540x960 padded by InputPadder), 32 iterations, in fp32 - plus the fp64 evaluation of the same arithmetic by the
oracle (the reference casts the feature maps and the lookup to fp32 internally, raft.py:191-193 / corr.py:50, so it
cannot be run in double as it is; oracle/ffraft_ref.py keeps fp64 for exactly this kind of noise study and is held
to the reference's fp32 outputs right here).

With the weights of the other fixtures (flow_head.conv2 damped x0.05) the 32-step recurrence is not contractive at this
size: the reference's own fp32 and fp64 runs drift apart by 0.45 px.  Here flow_head.conv2 is damped x0.01
(oracle.weights.det_tensor(..., flow_head_damp=0.01)): |flow| reaches 12.5 px and fp32 vs fp64 stay within 3e-4 px,
so a 1e-3 px parity bound means something.  Stored: flow_low (68x120) of both runs, flow_up sub-sampled, and the
per-iteration fp32-vs-fp64 spread.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_c5.py   (about 1 minute on 8 cores)
"""
import os
import sys
import zlib
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/core/models/ff-raft")
sys.dont_write_bytecode = True

from FF_RAFT_Core.raft import RAFT  # noqa: E402  (reference)

from oracle import ffraft_ref as orc  # noqa: E402
from oracle.weights import det_tensor  # noqa: E402

C5_DAMP = 0.01
torch.set_num_threads(8)


def main():
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                    MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
    net = RAFT(in_channels=256, small=False, dropout=0.0, alternate_corr=False, abandon_fnet=False,
               inside_fusion="parallel", fuse_cnet=True, cfg=cfg)
    net.load_state_dict({k: det_tensor("flow_net." + k, v.shape, flow_head_damp=C5_DAMP) for k, v in net.state_dict().items()},
                        strict=True)
    net.eval()
    image1, image2, mask1, mask2 = orc.shifted_pair(1, 544, 960, seed=3)
    i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
    with torch.no_grad():
        preds32 = net(i1, i2, m1, m2, iters=32)                         # train-mode return: all 32 predictions
        fl32, fu32 = net(i1, i2, m1, m2, iters=32, test_mode=True)
        sd = {"flow_net." + k: v for k, v in net.state_dict().items()}
        o32 = orc.ffraft_forward(sd, image1, image2, mask1, mask2, raft_iters=32)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        taps = {}
        fl64, fu64 = orc.ffraft_forward(sd64, *[t.double() for t in (image1, image2, mask1, mask2)], raft_iters=32,
                                        test_mode=True, taps=taps)
        preds64 = [it["flow_up"] for it in taps["iters"]]
    assert torch.equal(preds32[-1], fu32)
    print("oracle fp32 vs reference fp32, final prediction:", (o32[-1] - fu32).abs().max().item())
    spread = np.array([(a.double() - b).abs().max().item() for a, b in zip(preds32, preds64)])
    print("fp32 vs fp64 per iteration:", " ".join(f"{s:.1e}" for s in spread))
    print("|flow| max", fu64.abs().max().item())
    np.savez_compressed(
        os.path.join(HERE, "fwd_c5_544x960_b1_it32.npz"),
        in_crc=np.array([zlib.crc32(t.contiguous().numpy().tobytes()) for t in (image1, image2, mask1)], dtype=np.int64),
        flow_low_fp32=fl32.numpy(), flow_low_fp64=fl64.numpy(),
        flow_up_sub_fp32=fu32[:, :, ::4, ::4].contiguous().numpy(), flow_up_sub_fp64=fu64[:, :, ::4, ::4].contiguous().numpy(),
        spread_per_iter=spread, damp=np.array(C5_DAMP))


if __name__ == "__main__":
    main()
