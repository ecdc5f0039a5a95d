"""EPELoss / CPCL / MixLoss (core/models/ff-raft/losses/losses.py:18-130) as ONE fused HIP pass per
prediction: read pred + gt once, accumulate the loss in fp64 and write d(loss)/d(pred).
Same constructors, call signature `(flow_preds, flow_gt, valid, mask)` and `(loss, metrics)` return
as the reference; `build_losses` mirrors losses/__init__.py:3-11."""
import numpy as np
import torch
import torch.nn as nn

from . import _hip
from .ops import _p, _stream


def get_kernel(kernel_size, sigma):
    """losses.py:7-15 (host-side numpy, float64 -> float32) restated."""
    s3 = 3 * sigma
    xs = np.linspace(-s3, s3, kernel_size)
    x, y = np.meshgrid(xs, xs)
    gauss = 1 / (2 * np.pi * sigma ** 2) * np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return torch.FloatTensor((1 / gauss.sum()) * gauss).view(1, 1, kernel_size, kernel_size)


class _SequenceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, flow_gt, valid, mask, *preds):
        gamma, max_flow, ks, sigma, lam, use_mean, use_mask = cfg
        b, _, h, w = flow_gt.shape
        dev = flow_gt.device
        flow_gt, valid = flow_gt.contiguous(), valid.contiguous().float()
        vmap = torch.empty((b, h, w), dtype=torch.float32, device=dev)
        gconv = gsum = gk = None
        if use_mask:
            mask = mask.contiguous().float()
            gconv = torch.empty((b, h, w), dtype=torch.float32, device=dev)
            gsum = torch.zeros(1, dtype=torch.float64, device=dev)
            gk = get_kernel(ks, sigma).to(dev).contiguous()
        _hip.call("ff_loss_prepare", _p(flow_gt), _p(valid), _p(mask if use_mask else None), _p(gk), ks, float(max_flow),
                  _p(vmap), _p(gconv), _p(gsum), b, h, w, _stream())
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        n = len(preds)
        a_mean = 1.0 / (b * 2 * h * w) if use_mean else 0.0
        grads = []
        for i, pr in enumerate(preds):
            pr = pr.contiguous()
            g = torch.empty_like(pr) if ctx.needs_input_grad[4 + i] else None
            _hip.call("ff_loss_accumulate", _p(pr), _p(flow_gt), _p(vmap), _p(gconv), _p(gsum), a_mean, float(lam),
                      float(gamma ** (n - i - 1)), _p(g), _p(loss), b, h, w, _stream())
            grads.append(g)
        ctx.grads = grads
        ctx.vmap = vmap
        return loss.sum().float()     # a fresh 0-dim tensor, not a view: train.py:313-314 multiplies the loss IN PLACE (`loss *= world_size`)

    @staticmethod
    def backward(ctx, gout):
        scale = gout.float()
        return (None, None, None, None) + tuple(g * scale if g is not None else None for g in ctx.grads)


class _SeqLoss(nn.Module):
    use_mean, use_mask = True, False

    def __init__(self, gamma=0.8, max_flow=400, kernel_size=5, sigma=1.7, lamda=0.8):
        super().__init__()
        self.gamma, self.max_flow, self.kernel_size, self.sigma, self.lamda = gamma, max_flow, kernel_size, sigma, lamda

    def forward(self, flow_preds, flow_gt, valid, mask=None, *args):
        if not flow_gt.is_cuda:
            raise _hip.FocusFlowHipError("the fused loss runs on a HIP device only")
        cfg = (self.gamma, self.max_flow, self.kernel_size, self.sigma, self.lamda if self.use_mean and self.use_mask else 1.0,
               self.use_mean, self.use_mask)
        loss = _SequenceLossFn.apply(cfg, flow_gt, valid, mask, *flow_preds)
        with torch.no_grad():   # metrics (losses.py:39-45)
            b, _, h, w = flow_gt.shape
            vmap = torch.empty((b, h, w), dtype=torch.float32, device=flow_gt.device)
            _hip.call("ff_loss_prepare", _p(flow_gt.contiguous()), _p(valid.contiguous().float()), _p(None), _p(None), 1,
                      float(self.max_flow), _p(vmap), _p(None), _p(None), b, h, w, _stream())
            acc = torch.zeros(2, dtype=torch.float64, device=flow_gt.device)
            _hip.call("ff_epe_metric", _p(flow_preds[-1].detach().contiguous()), _p(flow_gt.contiguous()), _p(vmap), _p(acc),
                      b, h, w, _stream())
            s, n = acc.tolist()
        return loss, {"epe": s / n if n else float("nan"), "loss": loss.detach().item()}


class EPELoss(_SeqLoss):
    use_mean, use_mask = True, False

    def __init__(self, gamma=0.8, max_flow=400):
        super().__init__(gamma, max_flow)


class CPCL(_SeqLoss):
    use_mean, use_mask = False, True

    def __init__(self, gamma=0.8, max_flow=400, kernel_size=5, sigma=1.7):
        super().__init__(gamma, max_flow, kernel_size, sigma, 1.0)


class MixLoss(_SeqLoss):
    use_mean, use_mask = True, True


def build_losses(loss_type, gamma=0.8, max_flow=400, kernel_size=5, sigma=1.7, lamda=0.8, **kwargs):
    if loss_type == "EPELoss":
        return EPELoss(gamma, max_flow)
    if loss_type == "CPCL":
        return CPCL(gamma, max_flow, kernel_size, sigma)
    if loss_type == "MixLoss":
        return MixLoss(gamma, max_flow, kernel_size, sigma, lamda)
    raise ValueError(f'"loss_type":"{loss_type}" is not supported.')
