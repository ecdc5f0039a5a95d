#!/usr/bin/env python
"""Generate tests/golden/io_formats.npz with the reference's own readers / writers (authoring container only).

`core/utils/frame_utils.py` imports cv2 at module scope (only its KITTI PNG functions use it); with an empty
stand-in module its Middlebury .flo writer / reader and its PFM reader run unmodified.  Stored: the exact bytes
the reference's writeFlow produces for a known array, and what its readFlow / readPFM return for files written
here.  The KITTI 16-bit PNG functions need the real OpenCV and stay pinned by the format's definition.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_io.py
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/core")
sys.dont_write_bytecode = True
cv2 = types.ModuleType("cv2")
cv2.setNumThreads = lambda n: None
cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
sys.modules.setdefault("cv2", cv2)

from utils import frame_utils as ref  # noqa: E402  (reference)


def main():
    g = np.random.default_rng(7)
    flow = g.uniform(-50, 50, (5, 7, 2)).astype(np.float32)
    out = {"flow": flow}
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "a.flo")
        ref.writeFlow(fn, flow)
        out["flo_bytes"] = np.frombuffer(open(fn, "rb").read(), np.uint8)
        out["flo_read"] = ref.readFlow(fn)
        fn2 = os.path.join(d, "b.flo")
        ref.writeFlow(fn2, flow[..., 0], flow[..., 1])
        out["flo_bytes_uv"] = np.frombuffer(open(fn2, "rb").read(), np.uint8)
        img = g.uniform(-3, 3, (4, 6, 3)).astype(np.float32)
        pf = os.path.join(d, "c.pfm")
        with open(pf, "wb") as f:
            f.write(b"PF\n6 4\n-1.0\n")
            img.astype("<f4").tofile(f)
        out["pfm_bytes"] = np.frombuffer(open(pf, "rb").read(), np.uint8)
        out["pfm_read"] = np.ascontiguousarray(ref.readPFM(pf))
        out["pfm_read_gen"] = np.ascontiguousarray(ref.read_gen(pf))
        pg = os.path.join(d, "d.pfm")
        with open(pg, "wb") as f:
            f.write(b"Pf\n6 4\n1.0\n")
            img[..., 0].astype(">f4").tofile(f)
        out["pfm_gray_bytes"] = np.frombuffer(open(pg, "rb").read(), np.uint8)
        out["pfm_gray_read"] = np.ascontiguousarray(ref.readPFM(pg))
    np.savez_compressed(os.path.join(HERE, "io_formats.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
