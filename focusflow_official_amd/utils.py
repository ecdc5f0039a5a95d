"""Caller-side helpers the reference's evaluate.py uses around the model (core/utils/utils.py)."""
import torch.nn.functional as F


class InputPadder:
    """Replicate-pad (..., H, W) tensors so H and W become multiples of 8 (utils.py:7-24).
    mode 'sintel' centres the padding, anything else (KITTI) pads bottom/right-left as the reference does."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        pad_ht = (-self.ht) % 8
        pad_wd = (-self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        l, r, t, b = self._pad
        return x[..., t:ht - b, l:wd - r]
