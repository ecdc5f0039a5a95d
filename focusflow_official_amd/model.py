"""FF_RAFT_FUSION: the drop-in nn.Module (ff_raft.py:75-164) on the HIP path."""
import torch
import torch.nn as nn

from . import _hip, ops
from .raft_net import RAFT


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
        if modal != "point":
            raise NotImplementedError(f"MASK_MODAL={modal!r} is a 'next' row (SURVEY §8f-3); 'point' is built")
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

    def forward(self, image1, image2, mask1, mask2, raft_iters=12, flow_init=None, test_mode=False):
        b, c, h, w = image1.shape
        assert mask1.shape[1] == 1  # ff_raft.py:34
        if h % 8 or w % 8:
            raise ValueError("H and W must be multiples of 8 (pad with InputPadder as the reference's callers do)")
        # ff_raft.py:31-38 + :142-145 fused into one NCHW->NHWC4 pass per input;
        # 'point' mode ignores the caller's mask2 and uses a constant 255 plane.
        i1 = ops.prep_input(image1, b, h, w, image1)
        i2 = ops.prep_input(image2, b, h, w, image1)
        m1 = ops.prep_input(mask1, b, h, w, image1)
        m2 = ops.prep_input(None, b, h, w, image1, fill=255.0)
        return self.flow_net(i1, i2, m1, m2, iters=raft_iters, flow_init=flow_init, test_mode=test_mode)

    def freeze_self(self):
        if self.use_fusion == "parallel":
            self.flow_net.freeze_self(mode="parallel")
