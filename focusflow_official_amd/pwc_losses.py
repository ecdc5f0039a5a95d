"""FF-PWC multi-scale losses on the HIP path: EPELoss, CPCL, MixLoss and build_losses(cfg) with the reference's
constructor and call signatures (core/models/ff-pwcnet/losses/losses.py:19-261, losses/__init__.py).

    loss, metrics = criterion(flow_list, flow_gt[, mask], sparse=False)      # metrics = {'epe': ..., 'loss': ...}

Per pyramid level one launch builds the level's key-point weight map (bilinear mask resize, > 0, Gaussian) and one
launch does area-interpolation of the target, the end-point-error map, its weighted reduction and the gradient
w.r.t. the level's flow.  Dense ground truth only: the sparse (KITTI) variant raises NotImplementedError.
"""
import torch
import torch.nn as nn

from . import _hip, ops
from .model import gaussian_table
from .ops import _p, _stream


class _ScaleLoss(torch.autograd.Function):
    """loss contribution of one pyramid level; the kernel returns value and gradient together."""

    @staticmethod
    def forward(ctx, out, target, gmask, msum, w_plain, w_mask_num, zero_if_empty, over_batch, l1q, eps, q, sparse=False):
        b, _, h, w = out.shape
        oc = out.contiguous()
        grad = torch.empty_like(oc)
        loss = torch.zeros(1, dtype=torch.float64, device=out.device)
        _hip.call("ff_pwc_loss_scale_sparse" if sparse else "ff_pwc_loss_scale", _p(oc), _p(target), _p(gmask), _p(msum), float(w_plain), float(w_mask_num),
                  int(zero_if_empty), int(over_batch), int(l1q), float(eps), float(q), _p(grad), _p(loss), b, target.shape[2], target.shape[3],
                  h, w, _stream())
        ctx.save_for_backward(grad)
        return loss.sum().float()     # a fresh 0-dim tensor, not a view: train.py:313-314 multiplies the loss IN PLACE (`loss *= world_size`)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g,) + (None,) * 11


class EPELoss(nn.Module):
    """losses.py:19-86."""
    uses_mask = False
    mask_over_batch = False

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.loss_mode = cfg.TRAIN.LOSS_MODE
        self.weights = cfg.TRAIN.LOSS_WEIGHTS
        self.loss_q = cfg.TRAIN.LOSS_Q
        self.loss_epsilon = cfg.TRAIN.LOSS_EPSILON
        self._gauss = None

    # (w_plain, w_mask_num, zero_if_empty) of one level: EPE_map.sum() / batch  (:43)
    def _level_weights(self, weight, b, h, w):
        return weight / b, 0.0, 0

    def _mode(self):
        l1q = self.loss_mode != "pretrain"
        return int(l1q), float(self.loss_epsilon or 0.0), float(self.loss_q or 1.0)

    def _mask_map(self, mask, h, w):
        b, _, hh, ww = mask.shape
        if self._gauss is None or self._gauss.device != mask.device:
            self._gauss = gaussian_table(self.kernel_size, self.sigma).to(mask.device)
        gmask = torch.empty((b, h, w), dtype=torch.float32, device=mask.device)
        msum = torch.zeros(1, dtype=torch.float64, device=mask.device)
        _hip.call("ff_pwc_loss_mask", _p(mask.contiguous()), _p(self._gauss), self.kernel_size, _p(gmask), _p(msum), b, hh, ww, h, w,
                  _stream())
        return gmask, msum

    def realEPE(self, output, target, sparse=False):
        """Mean error of the finest flow bilinearly resized to the target's size (:78-81, :121-138, :216-233); sparse: over
        the pixels whose target is not exactly (0, 0) (:33-37)."""
        b, _, h, w = target.shape
        src = output.detach().permute(0, 2, 3, 1)                       # NHWC view (contiguous when it comes from FF_PWCNET)
        src = src if src.is_contiguous() or src.stride(3) == 1 else src.contiguous()
        up = torch.empty((b, 2, h, w), dtype=torch.float32, device=target.device)
        _hip.call("ff_resize_bilinear", _p(src), ops._ld(src), 2, src.shape[1], src.shape[2], _p(up), b, h, w, 1.0, 1.0, _stream())
        out2 = torch.zeros(2, dtype=torch.float64, device=target.device)
        l1q, eps, q = self._mode()
        _hip.call("ff_pwc_epe_mean_sparse" if sparse else "ff_pwc_epe_mean", _p(up), _p(target.contiguous()), l1q, eps, q, _p(out2), b, h, w,
                  _stream())
        return (out2[0] / out2[1]).float()

    # sparse ground truth: the plain term counts valid pixels only (EPELoss, :33-41); MixLoss overrides (:186-214)
    sparse_plain_valid_only = True

    def multiscaleEPE(self, network_output, target_flow, mask=None, sparse=False):
        if sparse and self.mask_over_batch:
            # CPCL with sparse=True indexes a (B,h,w) error map with the validity mask (1-D result) and then multiplies it
            # by a convolution of that same boolean mask (losses.py:100-114): the reference itself raises there
            raise NotImplementedError("CPCL with sparse ground truth fails in the reference too (losses.py:100-114)")
        if not isinstance(network_output, (tuple, list)):
            network_output = [network_output]
        assert len(self.weights) == len(network_output)
        ops._require_gpu(target_flow)
        target_flow = target_flow.contiguous()
        l1q, eps, q = self._mode()
        loss = 0
        for output, weight in zip(network_output, self.weights):
            b, _, h, w = output.shape
            gmask = msum = None
            if self.uses_mask:
                gmask, msum = self._mask_map(mask, h, w)
            wp, wm, zero = self._level_weights(weight, b, h, w)
            flag = self.sparse_plain_valid_only if sparse else self.mask_over_batch
            loss = loss + _ScaleLoss.apply(output, target_flow, gmask, msum, wp, wm, zero, flag, l1q, eps, q, bool(sparse))
        return loss

    def forward(self, output, target, *args, sparse=False):
        if self.uses_mask:
            mask, args = args[0], args[1:]
        else:
            mask = None
        if args:
            sparse = args[0]
        loss = self.multiscaleEPE(output, target, mask, sparse)
        return loss, {"epe": self.realEPE(output[0], target, sparse), "loss": loss.detach()}


class CPCL(EPELoss):
    """losses.py:89-164: the error map weighted by the Gaussian-smoothed key-point mask only.  The reference multiplies
    a (B,h,w) error map by a (B,1,h,w) mask (:114), which broadcasts to (B,B,h,w): every sample's errors meet every
    sample's mask.  Reproduced as it is (`mask_over_batch`)."""
    uses_mask = True
    mask_over_batch = True

    def __init__(self, cfg):
        super().__init__(cfg)
        self.kernel_size = cfg.TRAIN.LOSS_KERNEL_SIZE
        self.sigma = cfg.TRAIN.LOSS_SIGMA

    def _level_weights(self, weight, b, h, w):      # EPE_map.sum() / mask.sum() * (h*w)  (:119)
        return 0.0, weight * h * w, 0


class MixLoss(EPELoss):
    """losses.py:167-258: plain error sum + lamda * key-point-weighted term."""
    uses_mask = True
    sparse_plain_valid_only = False      # :204-214: EPE_map.sum() runs over every pixel, only the masked term drops invalid ones

    def __init__(self, cfg):
        super().__init__(cfg)
        self.kernel_size = cfg.TRAIN.LOSS_KERNEL_SIZE
        self.sigma = cfg.TRAIN.LOSS_SIGMA
        self.lamda = cfg.TRAIN.LOSS_LAMDA

    def _level_weights(self, weight, b, h, w):      # EPE_map.sum() + lamda * maskE.sum() / mask.sum() * (h*w)  (:208-214)
        return weight, weight * self.lamda * h * w, 1


def build_losses(cfg):
    loss_type = cfg.TRAIN.LOSS_TYPE
    if loss_type == "EPELoss":
        return EPELoss(cfg)
    if loss_type == "CPCL":
        return CPCL(cfg)
    if loss_type == "MixLoss":
        return MixLoss(cfg)
    raise ValueError(f'"loss_type":"{loss_type}" is not supported.')
