"""torch.autograd.Function wrappers: every forward AND backward is a sequence of
libfocusflow_hip launches.  PyTorch contributes the tape (which node feeds which)
and gradient accumulation into `.grad`, nothing else.

The dispatchers at the bottom (`conv`, `norm`, ...) pick the fused inference
launch when no gradient is being recorded and the autograd node otherwise.
"""
from typing import List, Optional

import os
import weakref

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU

Tensor = torch.Tensor


def _dense(t: Tensor) -> Tensor:
    """Gradients may arrive as arbitrary views; kernels want pixel-dense NHWC with ld % 4 == 0."""
    if t.stride(-1) == 1 and t.is_contiguous():
        return t
    return t.contiguous()


# ----------------------------------------------------------------------------
class GraphScope:
    """Weight-gradient bookkeeping of ONE recorded forward pass.

    A conv that is applied many times in one graph (the update block: 12 iterations; fnet: both frames) would hand
    autograd one weight-gradient tensor per application - per application a zero-fill, an un-pack launch and an
    accumulation add (about 1800 small launches per training step).  Inside a scope every application of a conv is
    wired to ONE `ParamGate` node per conv instead of to the parameters themselves: the wgrad kernel adds into one
    buffer per conv (it accumulates with atomics anyway), the applications return no parameter gradient, and the gate -
    which the autograd engine runs exactly once per backward pass, after every application the loss reaches has run -
    hands the total to the parameters THROUGH autograd.  Hooks (DDP's reducer, find_unused_parameters=False) therefore
    see exactly one gradient per parameter, also when the loss reaches only some of the applications, and a second
    backward over a retained graph starts from an empty buffer again."""

    _arena_hint = 0      # bytes of zeroed scratch the last pass asked for (weight-gradient buffers, norm-backward sums)

    def __init__(self, device=None):
        self.acc, self.gates = {}, {}
        self._amax_pool, self._amax_used, self._amax_ready = None, 0, None
        self._arena, self._arena_used, self._arena_req = None, 0, 0
        # data_ptr of a gradient tensor -> (word with the bits of its max|.|, shape, weak reference to the tensor): left by
        # the producer of the gradient (NormFn.backward), consumed - and removed - by the ConvFn.backward it flows into.
        # An entry counts only while its tensor is ALIVE at that address: a gradient nobody consumed (the norm of a frozen
        # branch whose conv takes no hint) may be freed and its address handed to another gradient of the same shape.
        self.amax_hint = {}
        # (data_ptr, numel) of a residual block's input -> the gradient of its skip connection: left by the block's last norm
        # (NormFn.backward, defer_res), added by the block's FIRST convolution inside its input-gradient launch (ConvFn.backward,
        # take_res: the epilogue's residual operand) - instead of autograd's own add over two full-size tensors
        self.res_grads = {}
        if device is not None and torch.device(device).type == "cuda":
            self._refill(device)       # on the stream that opens the pass (the main stream), before any side stream exists

    def _refill(self, device):
        """A fresh pool of zeroed max|.| words.  Backward nodes run on several streams (encoder / branch streams are
        replayed by autograd): every stream other than the filling one waits for the fill before its first atomicMax."""
        self._amax_pool, self._amax_used = torch.zeros(4096, dtype=torch.int32, device=device), 0
        if self._arena is None and GraphScope._arena_hint and _ZERO_ARENA:
            # ONE fill for the ~260 zeroed buffers a backward pass accumulates into, sized by what the previous pass used
            self._arena = torch.zeros(GraphScope._arena_hint, dtype=torch.uint8, device=device)
        self._amax_ready = torch.cuda.Event()
        self._amax_ready.record(torch.cuda.current_stream(device))
        self._amax_fill_stream = torch.cuda.current_stream(device)

    def amax_word(self, device):
        if self._amax_pool is None or self._amax_used >= self._amax_pool.numel():
            self._refill(device)
        cur = torch.cuda.current_stream(device)
        if cur != self._amax_fill_stream:
            cur.wait_event(self._amax_ready)
            self._amax_pool.record_stream(cur)
        self._amax_used += 1
        return self._amax_pool[self._amax_used - 1:self._amax_used]

    def zeros(self, shape, dtype, device):
        """A zeroed buffer for a backward kernel to accumulate into: carved from the pass's arena (filled once, when the
        pass opened) while it lasts, else a fresh torch.zeros."""
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        off = (self._arena_used + 255) // 256 * 256
        self._arena_req = (self._arena_req + 255) // 256 * 256 + nbytes
        GraphScope._arena_hint = max(GraphScope._arena_hint, self._arena_req + 256)
        if self._arena is None or self._arena.device != torch.device(device) or off + nbytes > self._arena.numel():
            return torch.zeros(shape, dtype=dtype, device=device)
        cur = torch.cuda.current_stream(device)
        if cur != self._amax_fill_stream:
            cur.wait_event(self._amax_ready)
            self._arena.record_stream(cur)
        self._arena_used = off + nbytes
        return self._arena[off:off + nbytes].view(dtype).view(shape)

    def put_hint(self, dx, word):
        self.amax_hint[dx.data_ptr()] = (word, tuple(dx.shape), weakref.ref(dx))

    def take_hint(self, dy):
        """The max|.| word of gradient `dy` if its producer left one for exactly this tensor, else None."""
        h = self.amax_hint.pop(dy.data_ptr(), None)
        if h is None:
            return None
        t = h[2]()
        if t is None or t.data_ptr() != dy.data_ptr() or h[1] != tuple(dy.shape):
            return None                # the producer's tensor is gone: the address was recycled
        return h[0]

    def gated(self, pc, params):
        """The aliases of pc's parameters that this pass's applications take as inputs (one gate per conv group)."""
        g = self.gates.get(pc)
        if g is None:
            it = iter(ParamGate.apply(self, pc, *[p for p in params if p is not None]))
            g = self.gates[pc] = [None if p is None else next(it) for p in params]
        return g


class ParamGate(torch.autograd.Function):
    """Identity on a conv group's parameters; its backward delivers the weight/bias gradients that the group's
    applications accumulated in `scope.acc[pc]` during this backward pass."""

    @staticmethod
    def forward(ctx, scope, pc, *params):
        ctx.scope, ctx.pc = scope, pc
        ctx.set_materialize_grads(False)
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *unused):
        pc = ctx.pc
        acc = ctx.scope.acc.pop(pc, None)
        grads: List[Optional[Tensor]] = [None, None]
        if acc is not None and _group_unpackable(pc) and all(ctx.needs_input_grad[2:]):
            return tuple(grads + [g for g in unpack_group(pc, acc[0], acc[1]) if g is not None])
        off, k = 0, 2                     # k: position among this node's inputs (None parameters were not passed)
        for j, cv in enumerate(pc.convs):
            co = cv.out_channels
            grads.append(pc.unpack_wgrad(acc[0], j, off) if acc is not None and ctx.needs_input_grad[k] else None)
            k += 1
            if cv.bias is not None and getattr(pc, "use_bias", True):
                grads.append(acc[1][off:off + co].clone() if acc is not None and ctx.needs_input_grad[k] else None)
                k += 1
            off += co
        return tuple(grads)


def _group_unpackable(pc) -> bool:
    return _UNPACK_GROUP and type(pc).__name__ == "PackedConv" and len(pc.convs) <= 4 and len(pc.cin_slices or ()) <= 4


def unpack_group(pc, dw: Tensor, db: Tensor) -> List[Optional[Tensor]]:
    """Packed weight-gradient rows dw [Cout][K] + bias gradient db [Cout] of a PackedConv -> the gradients of its members'
    parameters in the order of pc.params() (None where a member has no bias of its own): every member's OIHW gradient and
    bias gradient from ONE launch, as views of one buffer (ff_unpack_wgrad_group)."""
    if not _group_unpackable(pc):
        out, off = [], 0
        for j, cv in enumerate(pc.convs):
            out.append(pc.unpack_wgrad(dw, j, off))
            out.append(db[off:off + cv.out_channels].clone() if (cv.bias is not None and pc.use_bias) else None)
            off += cv.out_channels
        return out
    has_b = [cv.bias is not None and pc.use_bias for cv in pc.convs]
    couts = [cv.out_channels for cv in pc.convs]
    offs = [sum(couts[:j]) for j in range(len(couts))]
    cin_src = pc.convs[0].in_channels
    flat = ops.unpack_wgrad_group(dw, db, couts, offs, has_b, cin_src, pc.cin_slices, pc.kh, pc.kw, pc.cin_pad)
    out, at = [], 0
    for cv, co, hb in zip(pc.convs, couts, has_b):
        n = co * cin_src * pc.kh * pc.kw
        out.append(flat[at:at + n].view(co, cin_src, pc.kh, pc.kw))
        at += n
        if hb:
            out.append(flat[at:at + co])
            at += co
        else:
            out.append(None)
    return out


_scope: Optional[GraphScope] = None
_UNPACK_GROUP = True     # one gradient-unpacking launch per packed convolution (tests switch it off to compare)
_LOOKUP_BWD_ALL = True   # one lookup-backward launch per pass instead of one per iteration
_ZERO_ARENA = True       # one zero fill per pass for the backward's accumulation buffers
_AMAX_HINT = True        # the norm backward measures max|dx| for the conv it feeds
_RES_GRAD_FUSED = os.environ.get("FF_TRAIN_RES_GRAD", "1") != "0"      # A/B switch: a residual block's skip gradient added inside its first convolution's input-gradient launch


def res_grad_fused(pc) -> bool:
    """Whether a residual block that opens with convolution `pc` (stride 1: its input IS the skip connection) hands the skip
    connection's gradient to that convolution's input-gradient launch in this pass: conv(..., take_res=True) and
    NormFn(..., defer_res=True) must be given the same answer."""
    return _RES_GRAD_FUSED and _scope is not None and pc.stride == 1


def begin_graph(device=None) -> GraphScope:
    """Open a weight-gradient scope for the forward pass that follows (RAFT.forward when recording)."""
    global _scope
    _scope = GraphScope(device)
    return _scope


def end_graph():
    global _scope
    _scope = None


NFIX = 10  # non-tensor arguments of ConvFn.forward


class ConvFn(torch.autograd.Function):
    """y = act(conv(cat(xs)) + bias) * out_scale, or with a residual y = act(conv(cat(xs)) + bias + res) (the activation
    then follows the sum: the GRU gates over a pre-computed context share).  tensors = xs..., [res], (w_i, b_i)..."""

    @staticmethod
    def forward(ctx, pc, act, out_scale, nseg, has_res, pad_out, fill_tail, scope, stats_out, take_res, *tensors):
        xs = list(tensors[:nseg])
        assert not take_res or (res_grad_fused(pc) and nseg == 1)
        ctx.take_res = (xs[0].data_ptr(), xs[0].numel()) if take_res else None
        res = tensors[nseg] if has_res else None
        assert not (has_res and act != ACT_NONE and out_scale != 1.0), "residual + activation + out_scale: no such layer"
        w, b = pc.get()
        out = None
        if pad_out:  # allocate the channel-padded tensor; the caller owns channels >= Cout (filled by fill_tail)
            bsz, h, wd, _ = xs[0].shape
            ho = (h + 2 * pc.pad[0] - pc.kh) // pc.stride + 1
            wo = (wd + 2 * pc.pad[1] - pc.kw) // pc.stride + 1
            full = ops.empty_nhwc(bsz, ho, wo, (pc.cout + 3) // 4 * 4, xs[0])
            out = full[..., :pc.cout]
        y = ops.conv2d(xs, w, b, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, act=ACT_NONE if has_res else act, out=out, res=res,
                       act_res=act if has_res else ACT_NONE, out_scale=out_scale, w_fmt=pc.fmt, dilation=pc.dil,
                       want_stats=stats_out is not None, w_frag=pc.frag() if pc.dma_f32_ok(xs) else None)
        if stats_out is not None:      # per-sample {sum, sum of squares} of y for the InstanceNorm that follows (a constant
            y, st = y                  # of the graph: NormFn's backward differentiates through the statistics itself)
            stats_out.append(st)
        if pad_out:
            if fill_tail is not None:
                fill_tail(full)
            y = full
        ctx.pc, ctx.act, ctx.out_scale, ctx.nseg, ctx.has_res = pc, act, out_scale, nseg, has_res
        ctx.nparam = len(tensors) - nseg - (1 if has_res else 0)
        ctx.save_for_backward(*xs, y if act != ACT_NONE else None)
        ctx.scope = scope      # not None: the parameter inputs are this pass's ParamGate aliases
        ctx.hints = _scope     # the pass's max|.| hints are looked up (and consumed) by EVERY conv, gated or not
        return y

    @staticmethod
    def backward(ctx, dy):
        pc, act, nseg = ctx.pc, ctx.act, ctx.nseg
        saved = ctx.saved_tensors
        xs, y = list(saved[:nseg]), saved[nseg]
        dy = _dense(dy)
        scope = ctx.scope
        hint = ctx.hints.take_hint(dy) if ctx.hints is not None else None
        if hint is not None and ops.act_bwd_is_alias(dy, act, ctx.out_scale, pc.cout):
            g, amax = dy, hint             # conv -> norm: the norm's backward kernel has measured max|dy| already
        else:
            g, amax = ops.act_bwd(dy, y, act, ctx.out_scale, pc.cout, want_amax=True,    # (B,Ho,Wo,Cpad), zero padded
                                  amax=scope.amax_word(dy.device) if scope is not None else None)
        grads: List[Optional[Tensor]] = [None] * NFIX
        # input gradient: forward conv over g with flipped/transposed weights
        need_dx = any(ctx.needs_input_grad[NFIX + i] for i in range(nseg))
        dxs = [None] * nseg
        skip = ctx.hints.res_grads.pop(ctx.take_res, None) if (ctx.take_res is not None and ctx.hints is not None) else None
        if need_dx:
            wd, dfmt = pc.get_dgrad()
            cin_tot = sum(x.shape[3] for x in xs)
            b, h, w, _ = xs[0].shape
            gi = g
            if pc.stride == 2:
                gi = ops.dilate2(g, h + 2 * pc.pad[0] - pc.kh + 1, w + 2 * pc.pad[1] - pc.kw + 1)
            elif pc.stride != 1:
                raise NotImplementedError("stride > 2")
            d = pc.dil
            dx = ops.conv2d([gi], wd, None, cin_tot, pc.kh, pc.kw, 1, (d * (pc.kh - 1) - pc.pad[0], d * (pc.kw - 1) - pc.pad[1]),
                            w_fmt=dfmt, x_amax=amax if dfmt else None, dilation=d, w_frag=pc.frag_dgrad() if pc.stride == 1 else None,
                            res=skip)          # (+ the skip connection's gradient: the block input's total leaves this launch)
            off = 0
            for i, x in enumerate(xs):
                if ctx.needs_input_grad[NFIX + i]:
                    dxs[i] = dx[..., off:off + x.shape[3]]
                off += x.shape[3]
        grads += dxs
        if ctx.has_res:
            grads.append(dy if act == ACT_NONE else g[..., :pc.cout])  # y = act(conv + res): the pre-activation gradient
        # parameter gradients
        base = NFIX + nseg + (1 if ctx.has_res else 0)
        need_w = any(ctx.needs_input_grad[base:])
        dwp = db = None
        if need_w and scope is not None:   # shared weights: add into the conv's buffer, deliver with the last application
            acc = scope.acc.get(pc)
            if acc is None:
                kdim = pc.kh * pc.kw * sum(x.shape[3] for x in xs)
                z = scope.zeros((pc.cout * kdim + pc.cout,), torch.float32, g.device)
                acc = scope.acc[pc] = (z[:pc.cout * kdim].view(pc.cout, kdim), z[pc.cout * kdim:])
            ops.conv2d_wgrad(xs, g, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, g_amax=amax, want_db=True, dw=acc[0], db=acc[1],
                             dilation=pc.dil)
            return tuple(grads + [None] * ctx.nparam)      # the pass's ParamGate delivers the total
        elif need_w:   # weight + bias gradient in one launch (f16 matrix pipe unless the conv precision is fp32)
            dwp, db = ops.conv2d_wgrad(xs, g, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, g_amax=amax, want_db=True, dilation=pc.dil)
        off = 0
        for j, cv in enumerate(pc.convs):
            co = cv.out_channels
            gw = gb = None
            if ctx.needs_input_grad[base + 2 * j]:
                gw = pc.unpack_wgrad(dwp, j, off)
            if ctx.needs_input_grad[base + 2 * j + 1]:
                gb = db[off:off + co]
            grads += [gw, gb]
            off += co
        return tuple(grads)


class NormFn(torch.autograd.Function):
    """y = relu?(norm(x)) ; with res: y = relu(y + res).  Instance (per-sample) or batch statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, per_sample, fixed, eps, relu, stats, defer_res=False):
        y = ops.norm_apply(x, stats, per_sample, eps, gamma, beta, act=ACT_RELU if relu else ACT_NONE, res=res)
        ctx.meta = (per_sample, fixed, eps, relu, res is not None)
        ctx.scope = _scope
        # defer_res: `res` is the input of the residual block's first convolution (conv(..., take_res=True)), which adds d res
        # inside its own input-gradient launch
        assert not defer_res or (res is not None and _scope is not None)
        ctx.res_key = (res.data_ptr(), res.numel()) if defer_res else None
        ctx.save_for_backward(x, gamma, beta, y if res is not None else None, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        per_sample, fixed, eps, relu, has_res = ctx.meta
        x, gamma, beta, y, stats = ctx.saved_tensors
        scope = ctx.scope
        word = scope.amax_word(x.device) if (scope is not None and _AMAX_HINT) else None
        bst = scope.zeros((x.shape[0] if per_sample else 1, x.shape[3], 2), torch.float64, x.device) if scope is not None else None
        dx, dres, bst = ops.norm_bwd(x, _dense(dy), y, stats, per_sample, fixed, eps, gamma, beta, relu, has_res, amax=word, bstats=bst)
        if word is not None:
            scope.put_hint(dx, word)
        dgamma = dbeta = None
        if gamma is not None and ctx.needs_input_grad[1]:
            dgamma = bst[0, :, 1].float()
        if beta is not None and ctx.needs_input_grad[2]:
            dbeta = bst[0, :, 0].float()
        if ctx.res_key is not None and dres is not None and ctx.needs_input_grad[3]:
            scope.res_grads[ctx.res_key] = dres
            dres = None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None


class ActFn(torch.autograd.Function):
    """dst = act(src) on a channel slice (torch.split + tanh/relu of raft.py:205-207)."""

    @staticmethod
    def forward(ctx, x, act):
        y = ops.empty_nhwc(*x.shape, x)
        ops.act_copy(x, y, act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(_dense(dy), y, ctx.act, 1.0, y.shape[3]), None


class GruRhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, h):
        ctx.save_for_backward(r, h)
        return ops.gru_rh(r, h)

    @staticmethod
    def backward(ctx, drh):
        r, h = ctx.saved_tensors
        return ops.gru_rh_bwd(_dense(drh), r, h)


class GruBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, q, h):
        ctx.save_for_backward(z, q, h)
        return ops.gru_blend(z, q, h)

    @staticmethod
    def backward(ctx, dhn):
        z, q, h = ctx.saved_tensors
        return ops.gru_blend_bwd(_dense(dhn), z, q, h)


class UpsampleFn(torch.autograd.Function):
    """flow_up = convex_upsample(flow, mask); flow = flow_prev.detach() + delta, so d(delta) = d(flow)."""

    @staticmethod
    def forward(ctx, flow4, delta, up_mask):
        ctx.save_for_backward(flow4, up_mask)
        return ops.upsample_flow(flow4, up_mask)

    @staticmethod
    def backward(ctx, dout):
        flow4, up_mask = ctx.saved_tensors
        dflow, dmask = ops.upsample_flow_bwd(dout, flow4, up_mask)
        return None, dflow, dmask


# ---- CorrBlock -------------------------------------------------------------
class CorrBuildFn(torch.autograd.Function):
    """(fmap1, fmap2) -> token.  The tiled pyramid lives on `block` (block.pyr); lookups accumulate their gradients
    into block.grad_pyr (tiled fp32 planes - no per-iteration volume-sized autograd buffers), and this node - which
    autograd runs after every LookupFn because of the token edge - folds them down the pooling chain to d(volume) and
    contracts that with the feature maps (BmmBackward of corr.py:58).  fp16 pyramid storage is a straight-through
    rounding for the gradient, as a cast under autocast would be."""

    @staticmethod
    def forward(ctx, f1, f2, block, half):
        block.pyr = ops.corr_build(f1, f2, half)
        block.grad_pyr = None
        ctx.block = block
        ctx.save_for_backward(f1, f2)
        return torch.zeros(1, device=f1.device)

    @staticmethod
    def backward(ctx, dtoken):
        f1, f2 = ctx.saved_tensors
        blk = ctx.block
        b, h, w, _ = f1.shape
        pending, blk.pending = getattr(blk, "pending", None), None
        if pending:
            # every lookup of the pass in one launch (+ the pooling chain): LookupFn.backward only queues where that launch
            # is going to accept the pass (plane sizes and count checked up front), so a refusal here is a library change
            d0 = ops.corr_lookup_tiled_bwd_all([c for c, _ in pending], [d for _, d in pending], blk.pyr.h0, blk.pyr.w0)
            if d0 is None:
                if blk.grad_pyr is None:
                    blk.grad_pyr = ops.TiledPyramid.empty(blk.pyr.levels[0].shape[0], blk.pyr.h0, blk.pyr.w0, False, f1.device, zero=True)
                for c, d in pending:
                    ops.corr_lookup_tiled_bwd(blk.grad_pyr, c, d)
            else:
                df1, df2 = ops.corr_volume_bwd(d0.view(b, h * w, -1), f1, f2, tiled=True)
                return df1, df2, None, None
        gp = blk.grad_pyr
        if gp is None:
            return torch.zeros_like(f1), torch.zeros_like(f2), None, None
        ops.corr_pyramid_tiled_bwd(gp)
        blk.grad_pyr = None
        df1, df2 = ops.corr_volume_bwd(gp.levels[0].view(b, h * w, -1), f1, f2, tiled=True)
        return df1, df2, None, None


class LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, block, coords):
        ctx.block = block
        ctx.save_for_backward(coords)
        return ops.corr_lookup_tiled(block.pyr, coords)

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        blk = ctx.block
        # deferred: the gradient of the pyramid is a sum over the lookups and nothing reads it before CorrBuildFn.backward,
        # which scatters all of them in one launch (ops.corr_lookup_tiled_bwd_all) - where that launch exists for the
        # pass: a query's four planes must fit its LDS and it takes at most LOOKUP_BWD_ALL_MAX lookups.  Otherwise the
        # gradient is scattered right away and nothing is retained (large crops would keep every dout alive on top of
        # the zeroed gradient pyramid).
        pend = getattr(blk, "pending", None) or []
        if _LOOKUP_BWD_ALL and blk.grad_pyr is None and len(pend) < ops.LOOKUP_BWD_ALL_MAX and ops.lookup_bwd_all_fits(blk.pyr.h0, blk.pyr.w0):
            pend.append((coords, _dense(dout)))
            blk.pending = pend
        else:
            if blk.grad_pyr is None:
                blk.grad_pyr = ops.TiledPyramid.empty(blk.pyr.levels[0].shape[0], blk.pyr.h0, blk.pyr.w0, False, dout.device, zero=True)
            blk.pending = None
            for c, d in pend + [(coords, _dense(dout))]:       # (the queue is only non-empty when the pass has more lookups than one launch takes)
                ops.corr_lookup_tiled_bwd(blk.grad_pyr, c, d)
        return torch.zeros(1, device=dout.device), None, None


# ----------------------------------------------------------------------------
# SA / CA fusion units (parallel_fusion.py:14-73)
# ----------------------------------------------------------------------------
class ChanStatsFn(torch.autograd.Function):
    """(B,H,W,C) -> (B,H,W,4) = [mean over channels, max over channels, 0, 0]."""

    @staticmethod
    def forward(ctx, x):
        st, am = ops.chan_stats(x)
        ctx.save_for_backward(am)
        ctx.c = x.shape[3]
        return st

    @staticmethod
    def backward(ctx, g):
        (am,) = ctx.saved_tensors
        return ops.chan_stats_bwd(_dense(g), am, ctx.c)


class SpatialStatsFn(torch.autograd.Function):
    """(B,H,W,C) -> (2B,1,1,C): per-sample mean over pixels stacked on the per-sample max."""

    @staticmethod
    def forward(ctx, x):
        out, am = ops.spatial_stats(x)
        ctx.save_for_backward(am)
        ctx.hw = x.shape[1:3]
        return out

    @staticmethod
    def backward(ctx, g):
        (am,) = ctx.saved_tensors
        return ops.spatial_stats_bwd(_dense(g), am, *ctx.hw)


class ScaleAddFn(torch.autograd.Function):
    """out = s * v + q; s per pixel (mode 0) or per (sample, channel), avg-branch + max-branch (mode 1)."""

    @staticmethod
    def forward(ctx, v, s, q, mode):
        ctx.save_for_backward(v, s)
        ctx.mode = mode
        return ops.scale_add(v, s, q, mode)

    @staticmethod
    def backward(ctx, gout):
        v, s = ctx.saved_tensors
        gout = _dense(gout)
        gv, gs = ops.scale_add_bwd(gout, v, s, ctx.mode)
        return gv, gs, gout, None


# ----------------------------------------------------------------------------
# dispatchers
# ----------------------------------------------------------------------------
def recording(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def conv(pc, xs, act=ACT_NONE, res=None, out_scale=1.0, pad_out=False, fill_tail=None, want_stats=False, take_res=False):
    """Convolution through a PackedConv group.  pad_out: return the channel-padded tensor.  want_stats: -> (y, stats),
    the per-sample {sum, sum of squares} table of y (ops.conv2d: from the convolution's epilogue where it can).  take_res: the
    input is also the skip connection of the residual block this convolution opens (NormFn defer_res)."""
    if not isinstance(xs, (list, tuple)):
        xs = [xs]
    params = pc.params()
    if recording(*xs, res, *params):
        scope = None
        if _scope is not None and ops.w_format() and recording(*params):
            scope, params = _scope, _scope.gated(pc, params)
        args = list(xs) + ([res] if res is not None else []) + params
        holder = [] if want_stats else None
        y = ConvFn.apply(pc, act, out_scale, len(xs), res is not None, pad_out, fill_tail, scope, holder, take_res, *args)
        return (y, holder[0]) if want_stats else y
    if pad_out:
        b, h, w, _ = xs[0].shape
        ho = (h + 2 * pc.pad[0] - pc.kh) // pc.stride + 1
        wo = (w + 2 * pc.pad[1] - pc.kw) // pc.stride + 1
        full = ops.empty_nhwc(b, ho, wo, (pc.cout + 3) // 4 * 4, xs[0])
        pc(xs, act=act, res=res, out_scale=out_scale, out=full[..., :pc.cout])
        if fill_tail is not None:
            fill_tail(full)
        return full
    return pc(xs, act=act, res=res, out_scale=out_scale, want_stats=want_stats)


def chan_stats(x):
    return ChanStatsFn.apply(x) if recording(x) else ops.chan_stats(x)[0]


def spatial_stats(x):
    return SpatialStatsFn.apply(x) if recording(x) else ops.spatial_stats(x)[0]


def scale_add(v, s, q, mode):
    return ScaleAddFn.apply(v, s, q, mode) if recording(v, s, q) else ops.scale_add(v, s, q, mode)
