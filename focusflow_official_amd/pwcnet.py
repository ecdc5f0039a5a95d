"""FF_PWCNET on the HIP path (core/models/ff-pwcnet/PWCNet_Core/ff_pwcnet.py:113-434) — inference.

Module tree and state_dict keys follow the reference (`netExtractor.{netOne..netSix, mask_net*, fusion1..6}`,
decoders `netTwo..netSix` with `netUpflow/netUpfeat/netOne..netSix`, `netRefiner.netMain`); the
nn.Conv2d / nn.ConvTranspose2d objects only hold parameters.  Execution: NHWC fp32 tensors through
libfocusflow_hip — LeakyReLU in the conv epilogue, dilated convs in the refiner, transposed convs as
zero-dilation + flipped-weight conv, the 81-channel cost volume and backwarp kernels of pwc.hip.

DenseNet-style `torch.cat([new, old], 1)` (ff_pwcnet.py:331-337) never materialises: each decoder level
owns ONE wide buffer laid out in the reference's final channel order, every piece padded to a multiple of
4 channels ([netFiv 32 | netFou 64 | netThr 96 | netTwo 128 | netOne 128 | volume 81+3 | tenOne C | flow 2+2 |
upfeat 2+2]); a conv reads the suffix starting at its own offset and writes its output slot in front of
it.  Packed weights get zero columns at the pad channels.

Built: 'point' masks, every fusion type, any input size (sizes that are not multiples of 64 are bilinearly
resized first and the flow is resized / rescaled back, as the reference does; BASELINE config 4, 448x1024,
needs no resize).  Training: when gradients are recorded every step goes through an autograd Function whose backward is HIP as well
(conv / transposed conv / dilated conv gradients, cost volume, backwarp); torch.cat builds the DenseNet tensors.
"""
import os

import torch
import torch.nn as nn

from . import _hip, fn, ops, pwc
from .cce import FusionUnit, PackedConv
from .ops import ACT_LEAKY, ACT_NONE, _p, _stream

LEVEL_CH = [16, 32, 64, 96, 128, 196]
BACKWARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}
GROWTH = [("netFiv", 32), ("netFou", 64), ("netThr", 96), ("netTwo", 128), ("netOne", 128)]   # buffer order (front to back)


def _pad4(c):
    return (c + 3) // 4 * 4


# Inference buffers pad the 81-channel volume to 96 and the two 2-channel pieces (up-sampled flow / feature) to 16 each:
# every DenseNet input then has a multiple of 32 channels and takes the patch-stationary / block-uniform conv kernels
# instead of the generic im2col loader (which was 65 % of the forward).  The autograd path keeps the 84 / 4 / 4 layout.
VOLP, FLP = 96, 16


_DIRECT_DECONV = True     # ff_deconv4x4s2_small for netUpflow / netUpfeat

class _Packed:
    """Packed weights of one conv whose input is a padded-piece buffer: `pieces` = [(real, padded), ...]."""

    def __init__(self, conv, pieces, transposed=False):
        self.conv, self.pieces, self.transposed = conv, pieces, transposed
        self.key = None

    def get(self):
        cv = self.conv
        key = (ops.conv_precision(), cv.weight._version, cv.weight.data_ptr(), cv.bias._version)
        if key != self.key:
            w = cv.weight.detach()
            if self.transposed:   # ConvTranspose2d [Cin][Cout][k][k] -> equivalent forward conv [Cout][Cin][k][k], flipped
                w = w.permute(1, 0, 2, 3).flip(2, 3).contiguous()
            co, ci, kh, kw = w.shape
            cpad = sum(p for _, p in self.pieces)
            assert ci == sum(r for r, _ in self.pieces)
            wp = torch.zeros((co, cpad, kh, kw), dtype=torch.float32, device=w.device)
            src = dst = 0
            for real, padded in self.pieces:       # scatter real channels to their padded positions (load-time plumbing)
                wp[:, dst:dst + real] = w[:, src:src + real]
                src, dst = src + real, dst + padded
            packed = torch.empty((co, kh * kw * cpad), dtype=torch.float32, device=w.device)
            ops.pack_conv_weight(wp, packed, cpad, 0)
            self.fmt = 0 if (co <= 2 and kh == 3 and not self.transposed) else ops.w_format()   # small heads: conv_small.hip (fp32)
            self.w32 = packed if (self.transposed and co <= 2 and kh == 4) else None           # -> ops.deconv4x4s2_small
            self.w = ops.pack_split(packed) if self.fmt else packed
            self.b = cv.bias.detach()
            self.cout, self.k = co, kh
            self.key = key
        return self.w, self.b

    def __call__(self, x, act=ACT_NONE, pad=1, dilation=1, stride=1, out=None, res=None):
        w, b = self.get()
        return ops.conv2d([x], w, b, self.cout, self.k, self.k, stride, pad, act=act, out=out, res=res, w_fmt=self.fmt,
                          dilation=dilation)


class _TrainPacked(PackedConv):
    """PackedConv for the autograd path when the input is a padded-piece tensor (pieces = [(real, padded), ...]) and /
    or the module is a ConvTranspose2d (run as a stride-1 conv over the zero-dilated input with the flipped,
    transposed kernel).  The kernels see a "virtual" weight (Cout, sum(padded), k, k); gradients are mapped back."""

    def __init__(self, conv, pieces, transposed=False):
        self.convs = [conv]
        self.transposed = transposed
        w = conv.weight
        self.cout = w.shape[1] if transposed else w.shape[0]
        self.kh, self.kw = w.shape[2], w.shape[3]
        self.stride = 1 if transposed else conv.stride[0]
        self.pad = (self.kh - 1 - conv.padding[0], self.kw - 1 - conv.padding[1]) if transposed else tuple(conv.padding)
        self.dil = 1 if transposed else conv.dilation[0]
        self.cin = self.cin_pad = sum(p for _, p in pieces)
        idx, pos = [], 0
        for real, padded in pieces:
            idx += list(range(pos, pos + real))
            pos += padded
        assert len(idx) == (w.shape[0] if transposed else w.shape[1])
        self._idx = torch.tensor(idx, dtype=torch.long)
        self._key = self._dkey = None
        self.w = self.b = self.wd = None

    def _virtual(self):
        cv = self.convs[0]
        w = cv.weight.detach()
        if self.transposed:   # ConvTranspose2d [Cin][Cout][k][k] -> forward conv [Cout][Cin][k][k], flipped
            w = w.permute(1, 0, 2, 3).flip(2, 3)
        wp = torch.zeros((self.cout, self.cin_pad, self.kh, self.kw), dtype=torch.float32, device=w.device)
        wp[:, self._idx.to(w.device)] = w
        return wp

    def get(self):
        cv = self.convs[0]
        key = (ops.conv_precision(), cv.weight._version, cv.weight.data_ptr(), cv.bias._version)
        if key != self._key:
            wp = self._virtual()
            self.w = torch.empty((self.cout, self.kh * self.kw * self.cin_pad), dtype=torch.float32, device=wp.device)
            ops.pack_conv_weight(wp, self.w, self.cin_pad, 0)
            small = self.cout <= 2 and (self.kh, self.kw, self.stride, self.dil) == (3, 3, 1, 1) and self.pad == (1, 1)
            self.fmt = 0 if small else ops.w_format()
            if self.fmt:
                self.w = ops.pack_split(self.w)
            self.b = cv.bias.detach()
            self._key = key
        return self.w, self.b

    def get_dgrad(self):
        cv = self.convs[0]
        key = (ops.conv_precision(), cv.weight._version, cv.weight.data_ptr())
        if key != self._dkey:
            cout_pad = (self.cout + 3) // 4 * 4
            self.wd = torch.zeros((self.cin_pad, self.kh * self.kw * cout_pad), dtype=torch.float32, device=cv.weight.device)
            ops.pack_conv_weight_dgrad(self._virtual().contiguous(), self.wd, cout_pad, 0)
            self.dfmt = ops.w_format()
            if self.dfmt:
                self.wd = ops.pack_split(self.wd)
            self._dkey = key
        return self.wd, self.dfmt

    def unpack_wgrad(self, dwp, j, off):
        full = ops.unpack_conv_wgrad(dwp, self.cout, self.cin_pad, self.kh, self.kw, self.cin_pad, off)   # virtual layout
        g = full[:, self._idx.to(full.device)]
        return g.flip(2, 3).permute(1, 0, 2, 3).contiguous() if self.transposed else g.contiguous()


class _Dilate2(torch.autograd.Function):
    """Zero-dilation by 2 (the first half of a stride-2 transposed convolution); backward = the even samples."""

    @staticmethod
    def forward(ctx, x):
        b, h, w, _ = x.shape
        return ops.dilate2(x.contiguous(), 2 * h - 1, 2 * w - 1)

    @staticmethod
    def backward(ctx, g):
        return g[:, ::2, ::2, :].contiguous()


class _CostVolume84(torch.autograd.Function):
    """FunctionCorrelation written into an 84-channel tensor (81 + 3 zero pad channels: conv inputs come in groups of 4)."""

    @staticmethod
    def forward(ctx, one, two):
        one, two = one.contiguous(), two.contiguous()
        ctx.save_for_backward(one, two)
        b, h, w, _ = one.shape
        out = torch.zeros((b, h, w, 84), dtype=torch.float32, device=one.device)
        pwc._cv_fwd(one, two, out=out[..., :81])
        return out

    @staticmethod
    def backward(ctx, gout):
        one, two = ctx.saved_tensors
        gout = gout.contiguous()
        b, h, w, _ = gout.shape
        g81 = gout[..., :81]
        g_one = pwc._cv_bwd(g81, two) if ctx.needs_input_grad[0] else None
        g_two = None
        if ctx.needs_input_grad[1]:
            gt = ops.empty_nhwc(b, h, w, 81, gout)
            _hip.call("ff_pwc_gout_transpose", _p(g81), ops._ld(g81), _p(gt), 81, b, h, w, _stream())
            g_two = pwc._cv_bwd(gt, one)
        return g_one, g_two


def _stage(cin, c):
    return nn.Sequential(nn.Conv2d(cin, c, 3, 2, 1), nn.LeakyReLU(0.1), nn.Conv2d(c, c, 3, 1, 1), nn.LeakyReLU(0.1),
                         nn.Conv2d(c, c, 3, 1, 1), nn.LeakyReLU(0.1))


class Extractor(nn.Module):
    NAMES = ["netOne", "netTwo", "netThr", "netFou", "netFiv", "netSix"]

    def __init__(self, fusion_type):
        super().__init__()
        cin = 3
        self._packs = {}
        for lvl, name in enumerate(self.NAMES):
            c = LEVEL_CH[lvl]
            for prefix in ("", "mask_"):
                st = _stage(cin, c)
                setattr(self, prefix + name, st)
                self._packs[prefix + name] = [_Packed(st[0], [(cin, _pad4(cin))]), _Packed(st[2], [(c, c)]), _Packed(st[4], [(c, c)])]
            setattr(self, f"fusion{lvl + 1}", FusionUnit(c, fusion_type, lvl < 5))
            cin = c
        # autograd path: plain PackedConvs (the first conv of every branch reads a 4-channel NHWC4 image)
        self._tpacks = {k: [PackedConv([getattr(self, k)[2 * i]], 4 if (i == 0 and k.endswith("netOne")) else None) for i in range(3)]
                        for k in self._packs}

    def run_train(self, x, mask):
        feats = []
        for lvl, name in enumerate(self.NAMES):
            for pk in self._tpacks[name]:
                x = fn.conv(pk, x, act=ACT_LEAKY)
            for pk in self._tpacks["mask_" + name]:
                mask = fn.conv(pk, mask, act=ACT_LEAKY)
            mask, x = getattr(self, f"fusion{lvl + 1}").run(mask, x)
            feats.append(x)
        return feats

    def run(self, x, mask):
        feats = []
        for lvl, name in enumerate(self.NAMES):
            for i, pk in enumerate(self._packs[name]):
                x = pk(x, act=ACT_LEAKY, stride=2 if i == 0 else 1)
            for i, pk in enumerate(self._packs["mask_" + name]):
                mask = pk(mask, act=ACT_LEAKY, stride=2 if i == 0 else 1)
            mask, x = getattr(self, f"fusion{lvl + 1}").run(mask, x)
            feats.append(x)
        return feats


class Decoder(nn.Module):
    def __init__(self, level):
        super().__init__()
        self.level = level
        cur = {6: 81, 5: 81 + 128 + 4, 4: 81 + 96 + 4, 3: 81 + 64 + 4, 2: 81 + 32 + 4}
        c_one = {6: 0, 5: 128, 4: 96, 3: 64, 2: 32}[level]
        self.base = [(81, VOLP)] + ([(c_one, c_one), (2, FLP), (2, FLP)] if level < 6 else [])
        base_t = [(81, 84)] + ([(c_one, c_one), (2, 4), (2, 4)] if level < 6 else [])       # autograd path
        if level < 6:
            prev_c1 = {5: 0, 4: 128, 3: 96, 2: 64}[level]     # tenOne channels of the PREVIOUS (coarser) level
            self.netUpflow = nn.ConvTranspose2d(2, 2, 4, 2, 1)
            self.netUpfeat = nn.ConvTranspose2d(cur[level + 1] + 448, 2, 4, 2, 1)
            prev_base = [(81, VOLP)] + ([(prev_c1, prev_c1), (2, FLP), (2, FLP)] if level + 1 < 6 else [])
            prev_base_t = [(81, 84)] + ([(prev_c1, prev_c1), (2, 4), (2, 4)] if level + 1 < 6 else [])
            self._upflow = _Packed(self.netUpflow, [(2, 4)], transposed=True)
            self._upfeat = _Packed(self.netUpfeat, [(c, c) for _, c in GROWTH] + prev_base, transposed=True)
        c = cur[level]
        grown = []
        self._convs = []
        self._tconvs = []                          # autograd path: the 84 / 4 / 4 padded-piece layout, through ConvFn
        for name, co in reversed(GROWTH):          # netOne first
            conv = nn.Conv2d(c + sum(g for g, _ in grown), co, 3, 1, 1)
            setattr(self, name, nn.Sequential(conv, nn.LeakyReLU(0.1)))
            self._convs.append((name, _Packed(conv, grown + self.base)))
            self._tconvs.append(_TrainPacked(conv, grown + base_t))
            grown = [(co, co)] + grown
        self.netSix = nn.Sequential(nn.Conv2d(c + 448, 2, 3, 1, 1))
        self._six = _Packed(self.netSix[0], grown + self.base)
        self.width = 448 + sum(p for _, p in self.base)
        self._tsix = _TrainPacked(self.netSix[0], grown + base_t)
        if level < 6:
            self._tupflow = _TrainPacked(self.netUpflow, [(2, 4)], transposed=True)
            self._tupfeat = _TrainPacked(self.netUpfeat, [(c, c) for _, c in GROWTH] + prev_base_t, transposed=True)

    def run_train(self, one, two, prev):
        """Autograd twin of run(): torch.cat builds the DenseNet tensor (same padded-piece channel order)."""
        zero_tail = lambda full: full[..., 2:].zero_()      # noqa: E731  (2 real channels + 2 zero pads)
        if prev is None:
            feat = fn.ActFn.apply(_CostVolume84.apply(one, two), ACT_LEAKY)
        else:
            flow = fn.conv(self._tupflow, _Dilate2.apply(prev["tenFlow"]), pad_out=True, fill_tail=zero_tail)
            upfeat = fn.conv(self._tupfeat, _Dilate2.apply(prev["tenFeat"]), pad_out=True, fill_tail=zero_tail)
            warped = pwc.backwarp(two, flow, BACKWARP_SCALE[self.level])
            vol = fn.ActFn.apply(_CostVolume84.apply(one, warped), ACT_LEAKY)
            feat = torch.cat([vol, one, flow, upfeat], 3)
        for pk in self._tconvs:
            feat = torch.cat([fn.conv(pk, feat, act=ACT_LEAKY), feat], 3)
        return {"tenFlow": fn.conv(self._tsix, feat, pad_out=True, fill_tail=zero_tail), "tenFeat": feat}

    def run(self, one, two, prev):
        """-> dict(tenFlow (B,h,w,4 padded), tenFeat = the level buffer)."""
        b, h, w, _ = one.shape
        buf = torch.zeros((b, h, w, self.width), dtype=torch.float32, device=one.device)
        if prev is None:
            warped = two
        else:
            flow_slot = buf[..., 448 + VOLP + one.shape[3]: 448 + VOLP + one.shape[3] + FLP]
            feat_slot = buf[..., 448 + VOLP + one.shape[3] + FLP:]
            pf, pb = prev["tenFlow"], prev["tenFeat"]
            hp, wp = pf.shape[1], pf.shape[2]
            if _DIRECT_DECONV:     # ConvTranspose2d(., 2, 4, 2, 1) without the zero-dilated copy and the matrix tile
                for pk, src, slot in ((self._upflow, pf, flow_slot), (self._upfeat, pb, feat_slot)):
                    pk.get()
                    ops.deconv4x4s2_small(src, pk.w32, pk.b, 2, slot)
            else:
                self._upflow(ops.dilate2(pf, 2 * hp - 1, 2 * wp - 1), pad=2, out=flow_slot[..., :2])      # ConvTranspose2d
                self._upfeat(ops.dilate2(pb, 2 * hp - 1, 2 * wp - 1), pad=2, out=feat_slot[..., :2])
            warped = pwc.backwarp(two, flow_slot, BACKWARP_SCALE[self.level])
            ops.act_copy(one, buf[..., 448 + VOLP: 448 + VOLP + one.shape[3]], ACT_NONE)
        vol = buf[..., 448:448 + VOLP]
        pwc._cv_fwd(one, warped, out=vol[..., :81], act=ACT_LEAKY)      # leaky_relu(volume) on the way out; the pad channels stay 0
        off = 448
        for (name, pk), (_, co) in zip(self._convs, reversed(GROWTH)):
            pk(buf[..., off:], act=ACT_LEAKY, out=buf[..., off - co:off])
            off -= co
        flow4 = torch.zeros((b, h, w, 4), dtype=torch.float32, device=one.device)   # 2 flow channels + 2 zero pads
        self._six(buf, out=flow4[..., :2])
        return {"tenFlow": flow4, "tenFeat": buf}


class Refiner(nn.Module):
    def __init__(self):
        super().__init__()
        chans = [(565, 128), (128, 128), (128, 128), (128, 96), (96, 64), (64, 32), (32, 2)]
        self.dils = [1, 2, 4, 8, 16, 1, 1]
        layers = []
        for i, ((ci, co), d) in enumerate(zip(chans, self.dils)):
            layers.append(nn.Conv2d(ci, co, 3, 1, d, d))
            if i < 6:
                layers.append(nn.LeakyReLU(0.1))
        self.netMain = nn.Sequential(*layers)
        first = [(c, c) for _, c in GROWTH] + [(81, VOLP), (32, 32), (2, FLP), (2, FLP)]
        first_t = [(c, c) for _, c in GROWTH] + [(81, 84), (32, 32), (2, 4), (2, 4)]
        self._packs = [_Packed(self.netMain[0], first)] + [_Packed(self.netMain[2 * i], [(chans[i][0], chans[i][0])]) for i in range(1, 7)]
        self._tpacks = [_TrainPacked(self.netMain[0], first_t)] + [PackedConv([self.netMain[2 * i]]) for i in range(1, 7)]

    def run_train(self, feat, flow):
        x = feat
        for i, pk in enumerate(self._tpacks):
            if i < 6:
                x = fn.conv(pk, x, act=ACT_LEAKY)
            else:   # tenFlow + netRefiner(tenFeat), ff_pwcnet.py:424: the residual add rides in the conv epilogue
                return fn.conv(pk, x, res=flow[..., :2])

    def run(self, feat, flow):
        x = feat
        for i, (pk, d) in enumerate(zip(self._packs, self.dils)):
            if i < 6:
                x = pk(x, act=ACT_LEAKY, pad=d, dilation=d)
            else:
                out4 = torch.zeros_like(flow)
                pk(x, pad=d, dilation=d, res=flow[..., :2], out=out4[..., :2])   # tenFlow + netRefiner(tenFeat), ff_pwcnet.py:424
                return out4


class FF_PWCNET(nn.Module):
    """Same constructor and call as the reference: `model(im1, im2, mask1, mask2, test_mode=False)`, images
    (B,3,H,W) in [0,255] (NOT normalised), masks (B,1,H,W); returns 5 flows (1/4 ... 1/64 resolution, NCHW)
    or, in test_mode, the 1/4-resolution flow bilinearly resized to the input size."""

    def __init__(self, cfg, pretrain=None, load_pwcnet=None):
        super().__init__()
        _hip.load()
        if cfg.MODEL.FUSION != "parallel":
            raise NotImplementedError(f"FF_PWCNET only support parallel fusion, but got {cfg.MODEL.FUSION}")
        self.mask_modal = getattr(cfg.TRAIN, "MASK_MODAL", "point")
        if self.mask_modal not in ("point", "frame", "neighborG", "neighborE", "context"):
            raise ValueError(f"MASK_MODAL={self.mask_modal!r} is not one of point/frame/neighborG/neighborE/context")
        self._table = None
        self.fusion_type = cfg.MODEL.FUSION_TYPE
        self.cfg = cfg
        self.netExtractor = Extractor(self.fusion_type)
        self.netTwo, self.netThr, self.netFou, self.netFiv, self.netSix = (Decoder(l) for l in (2, 3, 4, 5, 6))
        self.netRefiner = Refiner()
        if pretrain is not None:
            self.load_state_dict(torch.load(pretrain), strict=True)
        if load_pwcnet is not None:
            self.load_state_dict(torch.load(load_pwcnet), strict=False)

    # Derived tensors (packed / split weights, folded BatchNorm) are cached per parameter version; the two entry points
    # below are where weights change wholesale, and `invalidate_packed()` is the explicit call for in-place `.data`
    # updates that leave the version counters alone.
    def invalidate_packed(self) -> int:
        from .cce import invalidate_packed
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

    def _nhwc4(self, t, b, h, w, like, fill=0.0):
        dst = ops.empty_nhwc(b, h, w, 4, like)
        _hip.call("ff_nchw_to_nhwc4", _p(t.contiguous() if t is not None else None), t.shape[1] if t is not None else 0,
                  fill, _p(dst), b, h, w, _stream())
        return dst

    def _resized4(self, t, b, h, w):
        dst = ops.empty_nhwc(b, h, w, 4, t)
        _hip.call("ff_resize_to_nhwc4", _p(t.contiguous()), t.shape[1], t.shape[2], t.shape[3], _p(dst), b, h, w, _stream())
        return dst

    def forward(self, tenOne, tenTwo, mask1, mask2, test_mode=False):
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        ops._require_gpu(tenOne)
        b, _, h0, w0 = tenOne.shape
        assert mask1.shape[1] == 1
        # preprocess (ff_pwcnet.py:391-403): bilinear resize to the next multiples of 64, a no-op when they already are
        h, w = (h0 + 63) // 64 * 64, (w0 + 63) // 64 * 64
        self.origin_H, self.origin_W, self.new_H, self.new_W = h0, w0, h, w
        resized = (h, w) != (h0, w0)
        if not resized:
            i1, i2 = self._nhwc4(tenOne, b, h, w, tenOne), self._nhwc4(tenTwo, b, h, w, tenOne)
        else:
            i1, i2 = self._resized4(tenOne, b, h, w), self._resized4(tenTwo, b, h, w)
        # init_mask (ff_pwcnet.py:61-110) on the PRE-PROCESSED inputs (:406-408); values stay in [0,255]
        modal = self.mask_modal
        if modal == "point":                                          # repeat to 3 channels; mask2 = 255
            m1 = self._resized4(mask1, b, h, w) if resized else self._nhwc4(mask1, b, h, w, tenOne)
            m2 = self._nhwc4(None, b, h, w, tenOne, fill=255.0)
        elif modal == "frame":                                        # :107-109: the masks are the frames
            m1, m2 = i1, i2
        else:
            from .model import MASK_MODES, ellipse_table, gaussian_table
            if self._table is None or self._table.device != tenOne.device:
                t = self.cfg.TRAIN
                tab = gaussian_table(t.KERNEL_SIZE, t.KERNEL_SIGMA) if modal == "neighborG" else ellipse_table(t.MASK_DILATE)
                self._table = tab.to(tenOne.device)
            mk = mask1.contiguous()
            if resized:                                               # (B,1,H,W) NCHW is NHWC with one channel
                mk = torch.empty((b, 1, h, w), dtype=torch.float32, device=tenOne.device)
                _hip.call("ff_resize_bilinear", _p(mask1.contiguous()), 1, 1, h0, w0, _p(mk), b, h, w, 1.0, 1.0, _stream())
            m1 = ops.mask_prepare(MASK_MODES[modal], mk, i1, self._table, raw=True, image_nhwc4=True)
            m2 = i2 if modal == "context" else self._nhwc4(None, b, h, w, tenOne, fill=255.0)
        decoders = ((6, self.netSix), (5, self.netFiv), (4, self.netFou), (3, self.netThr), (2, self.netTwo))
        est = None
        flows = []
        if train:   # every step through an autograd Function (fn.GraphScope shares the extractor's weight gradients)
            fn.begin_graph(i1.device)
            try:
                f1 = self.netExtractor.run_train(i1, m1)
                f2 = self.netExtractor.run_train(i2, m2)
                for level, dec in decoders:
                    est = dec.run_train(f1[level - 1], f2[level - 1], est)
                    if level == 2:
                        est["tenFlow"] = self.netRefiner.run_train(est["tenFeat"], est["tenFlow"])
                    flows.insert(0, est["tenFlow"])
            finally:
                fn.end_graph()
        else:
            # both frames through the extractor as ONE batch of 2B (no normalisation in it: the same arithmetic as the
            # reference's two calls, ff_pwcnet.py:436-437) - half the launches, grids twice as large at the tiny levels
            f12 = self.netExtractor.run(torch.cat([i1, i2], 0), torch.cat([m1, m2], 0))
            f1, f2 = [f[:b] for f in f12], [f[b:] for f in f12]
            for level, dec in decoders:
                est = dec.run(f1[level - 1], f2[level - 1], est)
                if level == 2:
                    est["tenFlow"] = self.netRefiner.run(est["tenFeat"], est["tenFlow"])
                flows.insert(0, est["tenFlow"])
        if test_mode:
            fl = est["tenFlow"][..., :2]
            out = torch.empty((b, 2, h0, w0), dtype=torch.float32, device=tenOne.device)     # back to the caller's size,
            _hip.call("ff_resize_bilinear", _p(fl), ops._ld(fl), 2, fl.shape[1], fl.shape[2], _p(out), b, h0, w0,
                      float(w0) / w, float(h0) / h, _stream())                                   # flow rescaled (:427-431)
            return out
        if train:   # NCHW views: the loss reads them as they are, autograd needs no extra node
            return [f[..., :2].permute(0, 3, 1, 2) for f in flows]
        return [ops.nhwc_to_nchw(f[..., :2]) for f in flows]
