#!/usr/bin/env python
"""Generate tests/golden/pwc_*.npz by RUNNING THE REFERENCE'S FF_PWCNET layers (authoring container only).

The reference module (`core/models/ff-pwcnet/PWCNet_Core/ff_pwcnet.py`) imports two things this image lacks:
`cv2` (only used by the non-'point' mask modes) and its CuPy-compiled `correlation` layer (needs cupy + a CUDA
device; its CPU path raises NotImplementedError, correlation.py:320-321).  To exercise everything else — the
Extractor / Decoder / Refiner wiring, the DenseNet concatenation order, the transposed convolutions, `backwarp`
(ff_pwcnet.py:27-47), the flow scales, `preprocess` and the test_mode resize — the two names are provided as
stand-in modules: an empty `cv2`, and a `correlation` whose FunctionCorrelation is the oracle's own
`cost_volume` (the formula of correlation.py:34-102).  `.cuda()` in backwarp is made a no-op.  The vectors therefore
pin the FF-PWC restatement in oracle/pwc_ref.py EXCEPT the cost-volume arithmetic itself, which the reference
cannot evaluate here and which stays pinned only by its definition.

Nothing of the reference is copied; only inputs' checksums, the state_dict spec and outputs are stored.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pwc.py
"""
import json
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
sys.path.insert(0, "/root/reference/core/models/ff-pwcnet")
sys.dont_write_bytecode = True

from oracle import pwc_ref  # noqa: E402
from oracle.weights import det_tensor  # noqa: E402

sys.modules.setdefault("cv2", types.ModuleType("cv2"))            # absent here; unused by MASK_MODAL='point'
corr_mod = types.ModuleType("correlation")
corr_mod.FunctionCorrelation = lambda tenOne, tenTwo: pwc_ref.cost_volume(tenOne, tenTwo)
pkg = types.ModuleType("correlation")
pkg.correlation = corr_mod
sys.modules["correlation"] = pkg
sys.modules["correlation.correlation"] = corr_mod
torch.Tensor.cuda = lambda self, *a, **k: self                    # backwarp() hard-codes .cuda() (ff_pwcnet.py:32)

from PWCNet_Core.ff_pwcnet import FF_PWCNET  # noqa: E402  (reference)

torch.manual_seed(0)
torch.set_num_threads(8)


def crc(t):
    return zlib.crc32(t.contiguous().numpy().tobytes())


def weights(net):
    """The deterministic, tamed weights of tests/test_hip_pwc.py::_pwc_weights (inputs are not normalised)."""
    sd = {}
    for k, v in net.state_dict().items():
        t = det_tensor("pwc." + k, v.shape)
        if k in ("netExtractor.netOne.0.weight", "netExtractor.mask_netOne.0.weight"):
            t = t / 255.0
        if ".netSix.0." in k or k.startswith("netRefiner.netMain.12") or "netUpf" in k:
            t = t * 0.1
        sd[k] = t
    return sd


def inputs(b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(b, 3, h // 4 + 4, w // 4 + 4, generator=g)
    i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
    i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
    m1 = (torch.rand(b, 1, h, w, generator=g) < 0.02).float() * 255
    return i1, i2, m1


def main():
    for ft in ("1x1conv", "concat"):
        cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE=ft))
        net = FF_PWCNET(cfg)
        sd = weights(net)
        net.load_state_dict(sd, strict=True)
        net.eval()
        with open(os.path.join(HERE, f"pwc_state_dict_spec_{ft}.json"), "w") as f:
            json.dump([[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()], f)
        out = {}
        for tag, (b, h, w) in (("128x192", (2, 128, 192)), ("100x180", (1, 100, 180))):   # the second one is pre-resized to 128x192
            i1, i2, m1 = inputs(b, h, w, seed=4 if tag == "128x192" else 5)
            with torch.no_grad():
                flows = net(i1, i2, m1, torch.zeros_like(m1))
                full = net(i1, i2, m1, torch.zeros_like(m1), test_mode=True)
            out[f"in_crc_{tag}"] = np.array([crc(i1), crc(i2), crc(m1)], dtype=np.int64)
            for lvl, fl in enumerate(flows):
                out[f"flow{lvl + 2}_{tag}"] = fl.numpy().astype(np.float32)
            out[f"full_{tag}"] = full.numpy().astype(np.float32)
            print(ft, tag, "levels", [tuple(f.shape) for f in flows], "max|flow|", float(full.abs().max()))
        np.savez_compressed(os.path.join(HERE, f"pwc_fwd_{ft}.npz"), **out)


if __name__ == "__main__":
    main()
