"""FF-PWC's native component on the HIP path: `FunctionCorrelation` (the reference's
`_FunctionCorrelation` autograd Function, correlation.py:276-381) and `backwarp`
(ff_pwcnet.py:27-47), NHWC fp32.  The full FF_PWCNET module is a later row."""
import torch

from . import _hip, ops
from .ops import _ld, _p, _stream, empty_nhwc


def _cv_fwd(one, two, out=None, act=0):
    """81-channel cost volume (act: the activation applied to it on the way out).  Coarse levels - a few 8 x 16 tiles with
    many channels - split the channel reduction over blocks (deterministic: fixed-order sum in a second launch)."""
    b, h, w, c = one.shape
    if out is None:
        out = empty_nhwc(b, h, w, 81, one)
    tiles, chunks = b * ((h + 7) // 8) * ((w + 15) // 16), (c + 15) // 16
    splits = min(chunks, 256 // tiles) if (tiles < 128 and chunks >= 4) else 1
    ws = torch.empty(splits * b * h * w * 81, dtype=torch.float32, device=one.device) if splits > 1 else None
    # (bench.py times these launches: label "pwc_costvolume", note = the launch's algorithmic bytes - both feature maps read once,
    # the 81 channels written once)
    ops._timed_call("pwc_costvolume", "ff_pwc_costvolume_fwd_ex", _p(one), _ld(one), _p(two), _ld(two), _p(out), _ld(out), b, h, w, c, act, _p(ws), splits,
                    _stream(), note=(b * h * w * (2 * c + 81) * 4, (b, h, w, c), splits))
    return out


def _cv_bwd(g, other):
    b, h, w, c = other.shape
    grad = empty_nhwc(b, h, w, c, other)
    _hip.call("ff_pwc_costvolume_bwd", _p(g), _ld(g), _p(other), _ld(other), _p(grad), c, b, h, w, c, _stream())
    return grad


class _FunctionCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, one, two):
        one, two = one.contiguous(), two.contiguous()
        ctx.save_for_backward(one, two)
        return _cv_fwd(one, two)

    @staticmethod
    def backward(ctx, gout):
        one, two = ctx.saved_tensors
        gout = gout.contiguous()
        b, h, w, _ = gout.shape
        g_one = _cv_bwd(gout, two) if ctx.needs_input_grad[0] else None
        g_two = None
        if ctx.needs_input_grad[1]:
            gt = empty_nhwc(b, h, w, 81, gout)
            _hip.call("ff_pwc_gout_transpose", _p(gout), 81, _p(gt), 81, b, h, w, _stream())
            g_two = _cv_bwd(gt, one)
        return g_one, g_two


def FunctionCorrelation(tenOne: torch.Tensor, tenTwo: torch.Tensor) -> torch.Tensor:
    """NHWC (B,H,W,C) x2 -> (B,H,W,81); channel (p+4)*9+(o+4), p = y-, o = x-displacement."""
    assert tenOne.shape == tenTwo.shape and tenOne.shape[3] % 4 == 0
    if not tenOne.is_cuda:  # the reference raises NotImplementedError on CPU as well (correlation.py:320-321)
        raise _hip.FocusFlowHipError("FunctionCorrelation runs on a HIP device only")
    return _FunctionCorrelation.apply(tenOne, tenTwo)


def _backwarp_fwd(tenInput, tenFlow, flow_scale):
    b, h, w, c = tenInput.shape
    out = empty_nhwc(b, h, w, c, tenInput)
    _hip.call("ff_pwc_backwarp", _p(tenInput), _ld(tenInput), _p(tenFlow), _ld(tenFlow), float(flow_scale), _p(out), c,
              b, h, w, c, _stream())
    return out


class _Backwarp(torch.autograd.Function):
    """grid_sample(bilinear, zeros, align_corners=False) x validity mask; the mask itself carries no gradient
    (ff_pwcnet.py:45 overwrites it with constants)."""

    @staticmethod
    def forward(ctx, tenInput, tenFlow, flow_scale):
        tenInput, tenFlow = tenInput.contiguous(), tenFlow.contiguous()
        ctx.save_for_backward(tenInput, tenFlow)
        ctx.scale = float(flow_scale)
        return _backwarp_fwd(tenInput, tenFlow, flow_scale)

    @staticmethod
    def backward(ctx, gout):
        tenInput, tenFlow = ctx.saved_tensors
        gout = gout.contiguous()
        b, h, w, c = tenInput.shape
        din = torch.zeros_like(tenInput) if ctx.needs_input_grad[0] else None
        dflow = torch.zeros_like(tenFlow) if ctx.needs_input_grad[1] else None      # channels >= 2 (padding) stay zero
        _hip.call("ff_pwc_backwarp_bwd", _p(tenInput), _ld(tenInput), _p(tenFlow), _ld(tenFlow), ctx.scale, _p(gout), _ld(gout),
                  _p(din), c, _p(dflow), tenFlow.shape[3], b, h, w, c, _stream())
        return din, dflow, None


def backwarp(tenInput: torch.Tensor, tenFlow: torch.Tensor, flow_scale: float = 1.0) -> torch.Tensor:
    """NHWC input (B,H,W,C), flow (B,H,W,>=2) [x,y] in pixels (times flow_scale) -> warped (B,H,W,C) with the
    validity mask applied."""
    if torch.is_grad_enabled() and (tenInput.requires_grad or tenFlow.requires_grad):
        return _Backwarp.apply(tenInput, tenFlow, flow_scale)
    return _backwarp_fwd(tenInput, tenFlow, flow_scale)
