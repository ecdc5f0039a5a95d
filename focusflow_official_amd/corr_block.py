"""CorrBlock on the HIP path (corr.py:12-60): all-pairs volume on the f16 matrix pipe (fp16-split operands, fp32-level
accuracy) with the 4-level average-pool pyramid written once from the accumulators in a tiled HBM layout, and the
radius-4 bilinear window lookup over that layout."""
import os

import torch

from . import fn, ops

PYRAMID_DTYPES = ("fp32", "fp16")
_MAX_PYRAMID_BYTES = int(os.environ.get("FF_MAX_PYRAMID_BYTES", str(3900 * 1000 * 1000)))      # (tests lower it to exercise the chunking)


class CorrBlock:
    """Same call protocol as the reference: build once per pair, call per iteration.

    fmap1/fmap2: NHWC (B, H8, W8, C) fp32.  ``__call__(coords)`` takes NHWC (B, H8, W8, 2) [x, y] coordinates and
    returns NHWC (B, H8, W8, levels*(2r+1)^2).  HBM layout: ops.TiledPyramid (128-byte 2-D tiles per plane).
    ``pyramid_dtype``: "fp32" (default; the reference's arithmetic) or "fp16" (storage only: every level is rounded to
    half after it has been computed in fp32 from the stored level below, as torch autocast would - BASELINE configs[4];
    the lookup still interpolates in fp32).  ``corr_pyramid`` gives the levels back as row-major fp32 planes."""

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 pyramid_dtype: str = None):
        if num_levels != 4 or radius != 4:
            raise NotImplementedError("the tiled CorrBlock kernels are built for 4 levels, radius 4 (all reference configs)")
        pyramid_dtype = pyramid_dtype or os.environ.get("FF_CORR_PYRAMID", "fp32")
        if pyramid_dtype not in PYRAMID_DTYPES:
            raise ValueError(f"pyramid_dtype must be one of {PYRAMID_DTYPES}")
        self.num_levels = num_levels
        self.radius = radius
        self.half = pyramid_dtype == "fp16"
        self.grad_pyr = None
        self._token = None
        fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
        # The lookup kernel addresses the four levels of a pyramid through ONE buffer resource (32-bit byte offsets): a batch
        # whose pyramid would pass 4 GB - configs[4] beyond 22 pairs: 177 MB of fp16 planes per pair - is built and looked up
        # in batch chunks, each with its own allocation (inference; a recorded pass keeps one pyramid for its backward).
        self._chunks = None
        b, h, w, _ = fmap1.shape
        per_pair = h * w * sum(ops.TiledPyramid.plane_elems(h, w, l, self.half) for l in range(4)) * (2 if self.half else 4) if fmap1.is_cuda else 0
        if per_pair * b >= _MAX_PYRAMID_BYTES and b > 1 and not fn.recording(fmap1, fmap2):
            per = max(1, _MAX_PYRAMID_BYTES // per_pair)
            self._chunks = [(lo, min(b, lo + per), CorrBlock(fmap1[lo:lo + per], fmap2[lo:lo + per], num_levels, radius, pyramid_dtype))
                            for lo in range(0, b, per)]
            self.pyr, self._nk, self._pairs, self._padded = None, num_levels * (2 * radius + 1) ** 2, b, None
            return
        if fn.recording(fmap1, fmap2):
            self._token = fn.CorrBuildFn.apply(fmap1, fmap2, self, self.half)      # sets self.pyr
        else:
            self.pyr: ops.TiledPyramid = ops.corr_build(fmap1, fmap2, self.half)
        # Inference: the lookup writes into a buffer whose channel count is padded to a multiple of 32 (324 -> 352,
        # pad channels zero once), so that convc1 takes the block-uniform loader (32-channel chunks) of the conv
        # kernel instead of the generic im2col one.  The buffer is reused by every iteration of this pair.
        self._nk = num_levels * (2 * radius + 1) ** 2
        self._pairs = fmap1.shape[0]
        self._padded = None

    @property
    def corr_pyramid(self):
        """The reference's attribute: [(B*Q, h_l, w_l) fp32 planes] (converted from the tiled storage on demand)."""
        if self._chunks is not None:
            return [torch.cat([blk.pyr.rowmajor(l) for _, _, blk in self._chunks], 0) for l in range(self.num_levels)]
        return [self.pyr.rowmajor(l) for l in range(self.num_levels)]

    def __call__(self, coords: torch.Tensor, want_taps: bool = False):
        if self._chunks is not None:
            assert not want_taps, "taps of a chunked CorrBlock: ask the chunks"
            if self._padded is None:
                b, h, w, _ = coords.shape
                self._padded = torch.zeros((b, h, w, (self._nk + 31) // 32 * 32), dtype=torch.float32, device=coords.device)
            for lo, hi, blk in self._chunks:
                ops.corr_lookup_tiled(blk.pyr, coords[lo:hi], out=self._padded[lo:hi][..., :self._nk])
            return self._padded
        if self._token is not None and not want_taps:
            return fn.LookupFn.apply(self._token, self, coords)
        if want_taps or torch.is_grad_enabled():
            return ops.corr_lookup_tiled(self.pyr, coords, want_taps)
        if self._padded is None:
            b, h, w, _ = coords.shape
            self._padded = torch.zeros((b, h, w, (self._nk + 31) // 32 * 32), dtype=torch.float32, device=coords.device)
        ops.corr_lookup_tiled(self.pyr, coords, out=self._padded[..., :self._nk])
        return self._padded
