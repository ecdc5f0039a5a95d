"""CPU oracle for FF-PWC's native component — TEST INFRASTRUCTURE.

Cost volume: PARITY UNPINNED.  The reference's own implementation (CuPy-compiled CUDA strings,
core/models/ff-pwcnet/PWCNet_Core/correlation.py:7-232) cannot run in this image (no cupy, no
CUDA device; its CPU path raises NotImplementedError, :320-321) and the reference holds no test
or golden vector for it.  `cost_volume` below restates the arithmetic those kernels spell out
(:34-102: channel = (p+4)*9+(o+4), s2o = ch%9-4 on x, s2p = ch/9-4 on y, zero padding by 4, mean
over C); its gradients come from autograd and agree with :104-166 / :168-232 by construction.

backwarp: restates ff_pwcnet.py:27-47 with the same ATen grid_sample call (minus `.cuda()`).
"""
import torch
import torch.nn.functional as F


def cost_volume(one: torch.Tensor, two: torch.Tensor) -> torch.Tensor:
    """NCHW (B,C,H,W) x2 -> (B,81,H,W)."""
    b, c, h, w = one.shape
    pad = F.pad(two, (4, 4, 4, 4))
    outs = []
    for p in range(-4, 5):          # y displacement (slow)
        for o in range(-4, 5):      # x displacement (fast)
            outs.append((one * pad[:, :, 4 + p:4 + p + h, 4 + o:4 + o + w]).sum(1, keepdim=True) / c)
    return torch.cat(outs, 1)


def backwarp(ten_input: torch.Tensor, ten_flow: torch.Tensor) -> torch.Tensor:
    b, _, h, w = ten_flow.shape
    hor = torch.linspace(-1.0 + (1.0 / w), 1.0 - (1.0 / w), w).view(1, 1, 1, -1).repeat(1, 1, h, 1)
    ver = torch.linspace(-1.0 + (1.0 / h), 1.0 - (1.0 / h), h).view(1, 1, -1, 1).repeat(1, 1, 1, w)
    grid = torch.cat([hor, ver], 1)
    flow = torch.cat([ten_flow[:, 0:1] / ((ten_input.shape[3] - 1.0) / 2.0),
                      ten_flow[:, 1:2] / ((ten_input.shape[2] - 1.0) / 2.0)], 1)
    inp = torch.cat([ten_input, ten_flow.new_ones(b, 1, h, w)], 1)
    out = F.grid_sample(inp, (grid + flow).permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = out[:, -1:].clone()
    mask[mask > 0.999] = 1.0
    mask[mask < 1.0] = 0.0
    return out[:, :-1] * mask
