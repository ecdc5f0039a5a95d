"""FF_RAFT_FUSION: the drop-in nn.Module (ff_raft.py:75-164) on the HIP path."""
import numpy as np
import torch
import torch.nn as nn

from . import _hip, ops
from .raft_net import RAFT

MASK_MODES = {"neighborG": 0, "neighborE": 1, "context": 2}


def gaussian_table(kernel_size, sigma):
    """ff_raft.py:13-21 (host-side numpy) restated."""
    s3 = 3 * sigma
    xs = np.linspace(-s3, s3, kernel_size)
    x, y = np.meshgrid(xs, xs)
    gauss = 1 / (2 * np.pi * sigma ** 2) * np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return torch.FloatTensor((1 / gauss.sum()) * gauss).contiguous()


def ellipse_table(k):
    """cv.getStructuringElement(cv.MORPH_ELLIPSE, (k, k)) as OpenCV 4.7 (requirements.txt pins
    opencv_python==4.7.0.72) computes it: row i spans columns c-dx .. c+dx with
    dx = round(c * sqrt((r*r - dy*dy) / (r*r))), r = c = k // 2, dy = i - r.  cv2 is absent from this
    image, so this is a restatement of the published algorithm (parity unpinned)."""
    r = c = k // 2
    el = np.zeros((k, k), np.float32)
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            el[i, max(c - dx, 0):min(c + dx + 1, k)] = 1.0
    if k == 1:
        el[:] = 1.0
    return torch.from_numpy(el).contiguous()


class FF_RAFT_FUSION(nn.Module):
    """Same constructor, attributes (`flow_net`, `fusion_layer`), call signature,
    state_dict keys and return values as the reference class.  Images are
    (B,3,H,W) fp32 in [0,255], masks (B,1,H,W) in {0..255}; H, W multiples of 8."""

    def __init__(self, pretrain=None, load_raft=None, use_fusion=None, fusion_channels=64, raft_small=False,
                 dropout=0., alternate_corr=False, abandon_fnet=False, fuse_cnet=False, freeze_flownet=False,
                 cfg=None):
        super().__init__()
        _hip.load()  # fail now, loudly, if the HIP library has not been built
        if use_fusion != "parallel":
            raise NotImplementedError(
                f"use_fusion={use_fusion!r}: only the 'parallel' (CCE) front-end is on the HIP path; "
                "'attention'/'conv' are selected by no shipped config (SURVEY §2.1 #8)")
        self.fusion_layer = None
        self.use_fusion = use_fusion
        self.freeze_flownet = freeze_flownet
        self.cfg = cfg
        modal = getattr(cfg.TRAIN, "MASK_MODAL", "point")
        if modal not in ("point", "frame") and modal not in MASK_MODES:
            raise ValueError(f"MASK_MODAL={modal!r} is not one of point/frame/neighborG/neighborE/context")
        self.mask_modal = modal
        self._table = None
        self.flow_net = RAFT(in_channels=fusion_channels, small=raft_small, dropout=dropout,
                             alternate_corr=alternate_corr, abandon_fnet=abandon_fnet,
                             inside_fusion="parallel", fuse_cnet=fuse_cnet, cfg=cfg)
        if pretrain is not None:
            self.load_state_dict(torch.load(pretrain), strict=True)
            print("Load pretrained model from {}".format(pretrain))
        if load_raft is not None:
            self.flow_net.load_model(load_raft, flag="all", strict=False)
            print("Load all flow net.")
        if self.freeze_flownet:
            self.freeze_self()
            print("freeze flow net.")

    # Derived tensors (packed / split weights, folded BatchNorm) are cached per parameter version; the two entry points
    # below are where weights change wholesale, and `invalidate_packed()` is the explicit call for in-place `.data`
    # updates that leave the version counters alone.
    def invalidate_packed(self) -> int:
        from .cce import invalidate_packed
        # new weights: what earlier forwards showed about the activation range no longer holds (ops: the always-on guard)
        self.flow_net._guard_hist, self.flow_net._guard_level, self.flow_net._exact_ctx = False, 0.0, False
        return invalidate_packed(self)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        # the default conv arithmetic splits a weight as f16(16 w) + residual: |w| must stay below 4094 (ff_common.h).
        # Checked where weights arrive wholesale (one reduction), not per step.
        big = [k for k, v in self.state_dict().items() if v.dim() == 4 and float(v.abs().max()) >= 4094.0] if ops.w_format() else []
        if big:
            raise ValueError(f"conv weights beyond the range of the f16x3 split format (|w| >= 4094): {big[:3]}; "
                             "use FF_CONV_PRECISION=fp32 for this checkpoint")
        return out

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def forward(self, image1, image2, mask1, mask2, raft_iters=12, flow_init=None, test_mode=False):
        b, c, h, w = image1.shape
        assert mask1.shape[1] == 1  # ff_raft.py:34
        if h % 8 or w % 8:
            raise ValueError("H and W must be multiples of 8 (pad with InputPadder as the reference's callers do)")
        # ff_raft.py:31-38 + :142-145 fused into one NCHW->NHWC4 pass per input;
        # 'point' mode ignores the caller's mask2 and uses a constant 255 plane.
        # (both frames - and both masks - into the halves of one buffer: the feature encoder takes them as one batch of 2B
        # without a torch.cat, ops.cat_batch)
        i12 = ops.empty_nhwc(2 * b, h, w, 4, image1)
        i1 = ops.prep_input(image1, b, h, w, image1, out=i12[:b])
        i2 = ops.prep_input(image2, b, h, w, image1, out=i12[b:])
        modal = self.mask_modal
        if modal == "point":
            m12 = ops.empty_nhwc(2 * b, h, w, 4, image1)
            m1 = ops.prep_input(mask1, b, h, w, image1, out=m12[:b])
            m2 = ops.prep_input(None, b, h, w, image1, fill=255.0, out=m12[b:])
        elif modal == "frame":                      # ff_raft.py:68-70: the masks are the frames themselves
            m1, m2 = i1, i2
        else:                                       # ff_raft.py:24-30, 40-66
            if self._table is None or self._table.device != image1.device:
                t = self.cfg.TRAIN
                tab = gaussian_table(t.KERNEL_SIZE, t.KERNEL_SIGMA) if modal == "neighborG" else ellipse_table(t.MASK_DILATE)
                self._table = tab.to(image1.device)
            m1 = ops.mask_prepare(MASK_MODES[modal], mask1, image1, self._table)
            m2 = i2 if modal == "context" else ops.prep_input(None, b, h, w, image1, fill=255.0)
        ops.guard_begin(image1.device)       # (raises if an EARLIER forward left the split formats' range: ops.guard_check)
        self.flow_net._guard_handled = False
        out = self.flow_net(i1, i2, m1, m2, iters=raft_iters, flow_init=flow_init, test_mode=test_mode)
        ops.guard_end("FF_RAFT_FUSION.forward", owner=self.flow_net, handled=self.flow_net._guard_handled)
        ops.check_range("FF_RAFT_FUSION.forward")      # debug mode FF_CHECK_RANGE=1 only (one host sync)
        return out

    def check_range(self):
        """The always-on range guard (ops.guard_*), now: waits for the forwards issued so far and raises FocusFlowHipError if
        one of them pushed an encoder output beyond the fp16-split formats' range (|x| >= 16376).  Without this call the
        same error is raised by the next forward."""
        ops.guard_check(sync=True)

    def freeze_self(self):
        if self.use_fusion == "parallel":
            self.flow_net.freeze_self(mode="parallel")
