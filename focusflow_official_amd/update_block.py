"""BasicUpdateBlock on the HIP path (update.py:79-146).

Parameter holders keep the reference's names; the forward is 10 convolution
launches + 4 gate kernels per iteration:
  convc1 -> convc2 ; convf1 -> convf2 ; conv(cat)            (motion encoder)
  [convz|convr] fused Cout=256 with sigmoid epilogue ; r*h ; convq with tanh
  epilogue on cat[r*h, x] (3 input segments, no concat buffer) ; blend   (x2)
  [flow_head.conv1|mask.0] fused Cout=512 with relu ; flow_head.conv2 ; mask.2 (x0.25)
"""
import os

import torch
import torch.nn as nn

from . import _hip, fn, ops
from .cce import PackedConv
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH


_SIDE_STREAM = True      # the motion encoder's flow branch on a second stream (inference; +0.7-1.2 %, docs/history.md)
_GRU_EPILOGUE = True    # r*h and the state blend in the conv epilogues (inference); tests switch it off to compare with ff_gru_rh / ff_gru_blend
# Inference: the activations BETWEEN the convolutions of the update block travel in the split-pair format (ops.SplitT) - the
# producers' epilogues write what the consumers' loaders would have made of fp32, the consumers take their patches by LDS-DMA
# (csrc/conv_dma.hip).  Same bits as the fp32 route.  FF_SPLIT_ACT=0: A/B switch back to fp32 tensors and conv_patch.hip.
_SPLIT_ACT = os.environ.get("FF_SPLIT_ACT", "1") != "0"
_GRU_PASS = os.environ.get("FF_GRU_PASS", "1") != "0"       # A/B switch: a SepConvGRU pass as one launch (z|r conv, r * h, q conv, blend)


def split_activations() -> bool:
    return (_SPLIT_ACT and _GRU_EPILOGUE and not torch.is_grad_enabled() and ops.w_format() in (_hip.W_F16X3, _hip.W_F16))
_side_streams = {}


def _side_stream(device):
    # one side stream per (device, stream the caller runs on): two forwards on two streams must not funnel their flow
    # branches through one queue
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self._zr = [PackedConv([self.convz1, self.convr1]), PackedConv([self.convz2, self.convr2])]
        self._q = [PackedConv([self.convq1]), PackedConv([self.convq2])]
        self.hidden_dim = hidden_dim
        # x = cat[inp, motion] (update.py:132) and `inp` - the context features - is the same in every iteration, so
        # its share of the six gate convolutions is computed once per forward (prepare) and enters the per-iteration
        # convolutions over [h, motion] as a pre-activation addend: a third of the GRU's matrix work leaves the loop.
        hm, ctx = [(0, hidden_dim), (hidden_dim + 128, c)], [(hidden_dim, hidden_dim + 128)]
        self._zr_hm = [PackedConv(g.convs, cin_slices=hm) for g in self._zr]
        self._q_hm = [PackedConv(g.convs, cin_slices=hm) for g in self._q]
        self._zr_ctx = [PackedConv(g.convs, cin_slices=ctx, use_bias=False) for g in self._zr]
        self._q_ctx = [PackedConv(g.convs, cin_slices=ctx, use_bias=False) for g in self._q]
        # the same in exact-fp32 rows (conv_mfma.hip): the always-on range guard's repair when the context features leave the
        # fp16-split formats' range (RAFT._exact_ctx)
        self._zr_ctx32 = [PackedConv(g.convs, cin_slices=ctx, use_bias=False, force_f32=True) for g in self._zr]
        self._q_ctx32 = [PackedConv(g.convs, cin_slices=ctx, use_bias=False, force_f32=True) for g in self._q]

    def prepare(self, inp, exact=False):
        """The context features' contribution to z|r and q of both passes, [(zr_pre, q_pre)] x 2.  Recorded passes too:
        autograd then sums the twelve pre-activation gradients that reach each share and runs the share's weight and
        input gradient ONCE (sum_t inp (x) g_t = inp (x) sum_t g_t).  exact (inference): the exact-fp32 rows - `inp` has left
        the split formats' range (RAFT._exact_ctx); the results are fp32 addends of the gates, which saturate."""
        assert inp.shape[3] == 128
        zr, q = (self._zr_ctx32, self._q_ctx32) if exact else (self._zr_ctx, self._q_ctx)
        return [(fn.conv(zc, inp), fn.conv(qc, inp)) for zc, qc in zip(zr, q)]

    def run_split(self, h, hs, motion, pre):
        """Inference on split-pair activations: h fp32 (the element-wise steps read it), hs = the same state as ops.SplitT
        (the convolutions read it), motion SplitT, pre = prepare(inp).  -> (h, hs).  update.py:45-60; the same sequence
        of operations as run()'s fused branch - bit-identical states."""
        c = self.hidden_dim
        if _GRU_PASS:
            # each pass as ONE launch (csrc/gru_pass.hip): r * h and z never leave the CU
            for d, (zr_conv, q_conv, (zr_pre, q_pre)) in enumerate(zip(self._zr_hm, self._q_hm, pre)):
                (_, bzr), (_, bq) = zr_conv.get(), q_conv.get()
                h, hs = ops.gru_pass(d, hs, motion, h, zr_pre, q_pre, zr_conv.frag(), q_conv.frag(), bzr, bq, zr_conv.fmt)
            return h, hs
        for zr_conv, q_conv, (zr_pre, q_pre) in zip(self._zr_hm, self._q_hm, pre):
            # [z | r * h]: z stays fp32 (the blend reads it), r * h leaves in the split-pair format (only the q convolution reads it)
            zr = zr_conv([hs, motion], res=zr_pre, act_res=ACT_SIGMOID, ep_rh=h, ep_split=c, y_split=c)
            # the new state twice: fp32 for the next element-wise steps, split-pair for the next convolutions
            h, hs = q_conv([ops.SplitT(zr[..., c:]), motion], res=q_pre, act_res=ACT_TANH, ep_blend=(zr[..., :c], h), y2_split=True)
        return h, hs

    def run(self, h, xs, pre=None):
        """h: (B,H,W,128); xs: list of NHWC segments forming x.  update.py:45-60.  pre: prepare(xs[0]) - then
        xs[0] itself is not read again."""
        c = self.hidden_dim
        if pre is not None and fn.recording(h, *xs, *[t for p in pre for t in p], *self.parameters()):
            for zr_conv, q_conv, (zr_pre, q_pre) in zip(self._zr_hm, self._q_hm, pre):
                zr = fn.conv(zr_conv, [h] + xs[1:], act=ACT_SIGMOID, res=zr_pre)
                rh = fn.GruRhFn.apply(zr[..., c:], h)
                q = fn.conv(q_conv, [rh] + xs[1:], act=ACT_TANH, res=q_pre)
                h = fn.GruBlendFn.apply(zr[..., :c], q, h)
            return h
        if pre is not None:
            # (not while a hipGraph is being captured: there the K splits of these short reductions pay more, and the
            # split partial sums cannot carry an epilogue)
            fused = _GRU_EPILOGUE and ops.w_format() in (_hip.W_F16X3, _hip.W_F16) and not torch.cuda.is_current_stream_capturing()
            for zr_conv, q_conv, (zr_pre, q_pre) in zip(self._zr_hm, self._q_hm, pre):
                if fused:
                    # the two element-wise steps ride in the epilogues of the convolutions that precede them (FFConvParams
                    # ep_mode; same roundings as ff_gru_rh / ff_gru_blend: bit-identical states): the z|r convolution
                    # writes [z | r * h], the q convolution writes (1 - z) h + z tanh(.) - 48 launches less per 12 iterations
                    zr = zr_conv([h] + xs[1:], res=zr_pre, act_res=ACT_SIGMOID, ep_rh=h, ep_split=c)
                    h = q_conv([zr[..., c:]] + xs[1:], res=q_pre, act_res=ACT_TANH, ep_blend=(zr[..., :c], h))
                    continue
                zr = zr_conv([h] + xs[1:], res=zr_pre, act_res=ACT_SIGMOID)     # sigmoid(conv([h, motion]) + b + pre)
                rh = ops.gru_rh(zr[..., c:], h)
                q = q_conv([rh] + xs[1:], res=q_pre, act_res=ACT_TANH)
                h = ops.gru_blend(zr[..., :c], q, h)
            return h
        for zr_conv, q_conv in zip(self._zr, self._q):
            zr = fn.conv(zr_conv, [h] + xs, act=ACT_SIGMOID)    # z = zr[..., :c], r = zr[..., c:]
            taped = fn.recording(zr, h)
            rh = fn.GruRhFn.apply(zr[..., c:], h) if taped else ops.gru_rh(zr[..., c:], h)
            q = fn.conv(q_conv, [rh] + xs, act=ACT_TANH)
            h = fn.GruBlendFn.apply(zr[..., :c], q, h) if taped else ops.gru_blend(zr[..., :c], q, h)
        return h


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)
        self._c1, self._c2 = PackedConv([self.convc1]), PackedConv([self.convc2])
        self._c1p = PackedConv([self.convc1], (cor_planes + 31) // 32 * 32)     # lookup output padded to 352 channels
        self._f1, self._f2 = PackedConv([self.convf1], 4), PackedConv([self.convf2])
        self._cv = PackedConv([self.conv])

    def run(self, flow4, corr, fill_flow, coords1=None):
        """flow4: (B,H,W,4) zero-padded flow.  Returns motion (B,H,W,128): 126 conv channels, and
        `fill_flow(motion)` writes the flow into channels 126:128 (torch.cat([out, flow]), update.py:97).
        coords1 given (inference, split_activations()): every tensor between the five convolutions is an ops.SplitT, the
        result too, and the last convolution writes the two flow channels itself (FF_EP_MOTION_TAIL: no fill launch)."""
        c1 = self._c1p if corr.shape[3] == self._c1p.cin_pad else self._c1
        sp = coords1 is not None
        ys = dict(y_split=True) if sp else {}

        def last(cor, flo):
            if not sp:
                return fn.conv(self._cv, [cor, flo], act=ACT_RELU, pad_out=True, fill_tail=fill_flow)
            b, h, w, _ = flow4.shape
            full = ops.empty_nhwc(b, h, w, 128, flow4)
            self._cv([cor, flo], act=ACT_RELU, out=full[..., :126], y_split=True, ep_motion_tail=coords1)
            return ops.SplitT(full)
        if _SIDE_STREAM and not ops.policy.single_stream and not torch.is_grad_enabled():
            # Inference: the flow branch (convf1 -> convf2) does not depend on the lookup and neither branch fills the
            # chip at 1/8 resolution (576 and 384 blocks on 1024 slots): run it on a second HIP stream beside
            # convc1 -> convc2.  It starts behind the lookup (event), so the lookup's own timing stays clean.
            main = torch.cuda.current_stream()
            side = _side_stream(flow4.device)
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(side):
                side.wait_event(fork)
                flo = self._f2(self._f1(flow4, act=ACT_RELU, **ys), act=ACT_RELU, **ys)
                join = torch.cuda.Event()
                join.record(side)
            flow4.record_stream(side)          # allocated on the main stream, read on the side stream
            cor = self._c2(c1(corr, act=ACT_RELU, **ys), act=ACT_RELU, **ys)
            main.wait_event(join)
            flo.record_stream(main)            # allocated on the side stream, read on the main stream
            return last(cor, flo)
        if sp:
            return last(self._c2(c1(corr, act=ACT_RELU, **ys), act=ACT_RELU, **ys), self._f2(self._f1(flow4, act=ACT_RELU, **ys), act=ACT_RELU, **ys))
        cor = fn.conv(self._c2, fn.conv(c1, corr, act=ACT_RELU), act=ACT_RELU)
        flo = fn.conv(self._f2, fn.conv(self._f1, flow4, act=ACT_RELU), act=ACT_RELU)
        return last(cor, flo)


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels, corr_radius, hidden_dim=128, input_dim=128):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))
        self._heads = PackedConv([self.flow_head.conv1, self.mask[0]])
        self._head1 = PackedConv([self.flow_head.conv1])      # flow head alone (RAFT.skip_unused_upsample)
        self._flow2 = PackedConv([self.flow_head.conv2])
        self._mask2 = PackedConv([self.mask[2]])

    def run(self, net, inp, corr, flow4, fill_flow, need_mask=True, gru_pre=None, defer_mask=False, coords_out=None):
        """-> (net, up_mask, delta_flow), all NHWC.  update.py:126-135.  need_mask=False (inference only, opt-in)
        leaves out the up-sampling mask head when the caller is going to discard it.  defer_mask (inference): up_mask
        comes back as the mask head's HIDDEN tensor (B,H,W,256) - the caller finishes it with upsample() below (the
        second mask convolution and the convex up-sampling as one launch).  coords_out = (coords1, flow4_next) (inference):
        the flow head's last convolution also takes the coordinate step coords1 += delta, flow4_next = coords1 - grid."""
        if isinstance(net, tuple):          # (h fp32, h split-pair): inference on split-pair activations (RAFT._loop_steps decides)
            motion = self.encoder.run(flow4, corr, fill_flow, coords1=net[2])
            h, hs = self.gru.run_split(net[0], net[1], motion, gru_pre)
            net, head_in = (h, hs, net[2]), hs
        else:
            motion = self.encoder.run(flow4, corr, fill_flow)
            net = head_in = self.gru.run(net, [inp, motion], gru_pre)
        if not need_mask:
            return net, None, fn.conv(self._flow2, self._head1(head_in, act=ACT_RELU) if isinstance(net, tuple) else fn.conv(self._head1, head_in, act=ACT_RELU))
        hid = self._heads(head_in, act=ACT_RELU) if isinstance(net, tuple) else fn.conv(self._heads, head_in, act=ACT_RELU)   # [flow-head 256 | mask-head 256]
        if coords_out is not None:
            delta = self._flow2(hid[..., :256], ep_coords=coords_out)
        else:
            delta = fn.conv(self._flow2, hid[..., :256])
        if defer_mask:
            return net, hid[..., 256:], delta
        up_mask = fn.conv(self._mask2, hid[..., 256:], out_scale=0.25)     # ".25 * self.mask(net)"
        return net, up_mask, delta

    def upsample(self, mask_hidden, flow4):
        """run(defer_mask=True)'s hidden tensor + the flow -> flow_up (B,2,8H,8W): ops.mask_upsample."""
        w, b = self._mask2.get()
        # re-arranged once per repack of the rows (PackedConv._gen: prepack rewrites the rows in place and a `.data` write
        # followed by cce.invalidate_packed repacks under an unchanged parameter key - neither may reuse the old image)
        gen = (self._mask2._gen, w.data_ptr())
        if getattr(self, "_mask2_stage_of", None) != gen:
            self._mask2_stage, self._mask2_stage_of = ops.mask_upsample_pack(w), gen
        return ops.mask_upsample(mask_hidden, self._mask2_stage, self._mask2.fmt, b, flow4, 0.25)

    def freeze_self(self, mode):
        if mode == "parallel":  # update.py:137-146
            for p in self.encoder.parameters():
                p.requires_grad = False
            for p in self.gru.parameters():
                p.requires_grad = False
            for p in self.flow_head.parameters():
                p.requires_grad = True
