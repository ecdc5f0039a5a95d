"""CorrBlock on the HIP path (corr.py:12-60): all-pairs volume on the fp32 matrix
pipe, 4-level average-pool pyramid, radius-4 bilinear window lookup."""
from typing import List

import torch

from . import fn, ops


class CorrBlock:
    """Same call protocol as the reference: build once per pair, call per iteration.

    fmap1/fmap2: NHWC (B, H8, W8, C) fp32.  ``__call__(coords)`` takes NHWC
    (B, H8, W8, 2) [x, y] coordinates and returns NHWC (B, H8, W8, levels*(2r+1)^2).
    HBM layout: level l is [B*Q][h_l][w_l] row-major fp32 planes (Q = H8*W8).
    """

    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4):
        if num_levels != 4:
            raise NotImplementedError("the pyramid kernel builds exactly 4 levels (all reference configs)")
        self.num_levels = num_levels
        self.radius = radius
        b, h, w, _ = fmap1.shape
        self.grad_levels = None
        self._token = None
        fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
        if fn.recording(fmap1, fmap2):
            vol = fn.CorrVolumeFn.apply(fmap1, fmap2)
            self._token = fn.PyramidFn.apply(vol, self, h, w)      # sets self.corr_pyramid
        else:
            self.corr_pyramid: List[torch.Tensor] = ops.corr_pyramid(ops.corr_volume(fmap1, fmap2), h, w)
        # Inference: the lookup writes into a buffer whose channel count is padded to a multiple of 32 (324 -> 352,
        # pad channels zero once), so that convc1 takes the block-uniform loader (32-channel chunks) of the conv
        # kernel instead of the generic im2col one.  The buffer is reused by every iteration of this pair.
        self._nk = num_levels * (2 * radius + 1) ** 2
        self._padded = None

    def __call__(self, coords: torch.Tensor, want_taps: bool = False):
        if self._token is not None and not want_taps:
            return fn.LookupFn.apply(self._token, self, coords)
        if want_taps or torch.is_grad_enabled():
            return ops.corr_lookup(self.corr_pyramid, coords, self.radius, want_taps)
        if self._padded is None:
            b, h, w, _ = coords.shape
            self._padded = torch.zeros((b, h, w, (self._nk + 31) // 32 * 32), dtype=torch.float32, device=coords.device)
        ops.corr_lookup(self.corr_pyramid, coords, self.radius, out=self._padded[..., :self._nk])
        return self._padded
