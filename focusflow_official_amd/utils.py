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


def forward_interpolate(flow):
    """Warm start for the next frame pair (utils.py:26-54): push every flow vector to where it lands,
    then fill the regular grid from the nearest landed vector (scipy.griddata 'nearest').  (2,H,W) -> (2,H,W), CPU."""
    import numpy as np
    import torch
    from scipy import interpolate
    f = flow.detach().cpu().numpy()
    ht, wd = f.shape[1:]
    x0, y0 = np.meshgrid(np.arange(wd), np.arange(ht))
    x1, y1 = (x0 + f[0]).reshape(-1), (y0 + f[1]).reshape(-1)
    keep = (x1 > 0) & (x1 < wd) & (y1 > 0) & (y1 < ht)
    pts = (x1[keep], y1[keep])
    out = [interpolate.griddata(pts, f[c].reshape(-1)[keep], (x0, y0), method="nearest", fill_value=0) for c in (0, 1)]
    return torch.from_numpy(np.stack(out, axis=0)).float()
