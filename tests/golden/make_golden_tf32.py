#!/usr/bin/env python
"""What the reference's OWN GPU arithmetic costs in flow accuracy (authoring container, CPU): every shipped config sets
ALLOW_TF32: true (config/experiment/ffraft_chairs_orb.yaml:7) and common.py:25-27 turns TF32 on for cudnn and matmul, so
on a GPU the reference reads every convolution / correlation operand with 10 mantissa bits.  The reference's CPU path
cannot do that, so this script evaluates the pinned oracle (bit-identical to the reference in fp32,
tests/test_oracle_golden.py) with both operands of every convolution and of the correlation matmul rounded to TF32
(oracle.ffraft_ref.round_tf32) on the BASELINE configs[0] / configs[1] inputs, and stores the end-point error against the
reference's fp32 fixture.  tests/test_hip_parity.py holds the HIP path's reduced-precision mode (single-term f16 operands:
the same 10 mantissa bits) to a small multiple of these numbers; bench.py reports the same measure.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tf32.py   (about 20 s on 8 cores)
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


def epe(a, b):
    return torch.sqrt(((a - b) ** 2).sum(1))


def main():
    with open(os.path.join(HERE, "state_dict_spec.json")) as f:
        sd = {k: det_tensor(k, s) for k, s, _ in json.load(f)}
    g = np.load(os.path.join(HERE, "fwd_shift_384x512_b1_it12.npz"))
    inp = orc.shifted_pair(1, 384, 512, seed=6)
    with torch.no_grad():
        fl32, fu32 = orc.ffraft_forward(sd, *inp, raft_iters=12, test_mode=True)
        assert np.abs(fl32.numpy() - g["flow_low"]).max() == 0.0, "the oracle no longer reproduces the reference fixture"
        orc.OPERAND_ROUND = orc.round_tf32
        flt, fut = orc.ffraft_forward(sd, *inp, raft_iters=12, test_mode=True)
        orc.OPERAND_ROUND = None
    e_up, e_low = epe(fut, fu32), epe(flt, fl32)
    out = {"input": "oracle.shifted_pair(1, 384, 512, seed=6), raft_iters=12 (fwd_shift_384x512_b1_it12)",
           "arithmetic": "conv + correlation operands rounded to TF32 (10 mantissa bits, RNE), fp32 accumulation",
           "epe_up_mean_px": float(e_up.mean()), "epe_up_max_px": float(e_up.max()),
           "epe_low_mean_px": float(e_low.mean()), "epe_low_max_px": float(e_low.max()),
           "flow_up_absmax_px": float(fu32.abs().max())}
    with open(os.path.join(HERE, "tf32_epe_384x512.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
