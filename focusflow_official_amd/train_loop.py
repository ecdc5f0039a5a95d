"""The recorded (training) update loop as ONE autograd node (raft.py:218-231, update.py:45-60 / :89-97 / :121-135).

The per-operation tape of fn.py gave the twelve iterations ~190 autograd nodes each; their backward was ~1000 small
launches of gradient accumulation (`add`), zero fills and per-application weight gradients.  Here the forward of all
iterations writes what the backward needs into STACKED buffers [T * B][H][W][C] (iteration-major), and the backward is a
fixed sequence of libfocusflow_hip launches:

  * the input-gradient chain of iteration T-1 ... 0 (eleven convolutions per iteration), the gate derivatives and every
    gradient sum between them in three element-wise kernels per iteration (csrc/train_ops.hip) - no accumulation launch;
  * ONE weight-gradient launch per convolution for all T iterations: the stacks are ordinary batches of T * B images, so
    `ff_conv2d_wgrad` contracts over all of them at once (13 launches instead of 156);
  * the lookup gradients of all iterations scattered by one launch, then the two volume contractions (fn.CorrBuildFn's job
    in the per-operation tape).

Parameter gradients leave this node once per parameter (DDP's reducer sees one gradient each).
"""
import os
from typing import List, Optional

import torch

from . import _hip, fn, ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH

Tensor = torch.Tensor

ENABLED = os.environ.get("FF_TRAIN_LOOP", "1") != "0"      # A/B switch: 0 = the per-operation tape of fn.py
# A/B switch: the recorded forward on the inference kernels (split-pair activations, conv_dma.hip, one launch per SepConvGRU pass with
# the gates stored for the backward, mask convolution + up-sampling as one kernel); 0 = conv by conv on fp32 tensors
FUSED_FWD = os.environ.get("FF_TRAIN_FUSED_FWD", "1") != "0"
WCHUNK = int(os.environ.get("FF_TRAIN_WCHUNK", "4"))       # iterations per weight-gradient slab (A/B: 12 = one launch per convolution, at the end)
WGRAD_SIDE_STREAM = os.environ.get("FF_TRAIN_WGRAD_STREAM", "1") != "0"      # A/B switch: the slabs on a side stream beside the input-gradient chain
_side_streams = {}


# The parameter gradients leave through a node of their own (LoopParamGate) that the model creates BEFORE the encoders run: the
# autograd engine serves ready nodes latest-created first, so that node's backward - which joins the side stream and un-packs the
# sums - runs after the encoders' backward has been issued instead of stalling the main stream behind the last slab of weight
# gradients (a millisecond of an idle main stream per step in the kernel trace).  Single-process training only: under DDP the update
# block's parameters would then be the LAST gradients to reach their bucket.
DEFER_PARAM_GRADS = os.environ.get("FF_TRAIN_DEFER_WGRAD", "1") != "0"


def defer_param_grads() -> bool:
    return DEFER_PARAM_GRADS and WGRAD_SIDE_STREAM and not (torch.distributed.is_available() and torch.distributed.is_initialized())


def _unpack_param_grads(convs, acc, need_of):
    """acc[i]: (dw, db) sums of convs[i] or None -> the gradients in the order of loop_params(); need_of(j): parameter j wanted."""
    grads, pos = [], 0
    for pc, a_ in zip(convs, acc):
        n = len(pc.params())
        if a_ is None:
            grads += [None] * n
        else:
            gl = fn.unpack_group(pc, a_[0], a_[1])
            grads += [gv if need_of(pos + i) else None for i, gv in enumerate(gl)]
        pos += n
    return grads


class LoopParamGate(torch.autograd.Function):
    """(*loop_params) -> a one-element token that UpdateLoopFn takes as an input.  Its backward receives nothing through the
    token; it picks up what UpdateLoopFn.backward left in `box` (the per-convolution gradient sums and the event behind the last
    weight-gradient launch), waits for that event and returns the parameters' gradients."""

    @staticmethod
    def forward(ctx, box, *params):
        ctx.box = box
        ctx.set_materialize_grads(False)
        return params[0].new_empty(1)

    @staticmethod
    def backward(ctx, _token):
        job = ctx.box.pop("job", None)
        n = len(ctx.needs_input_grad) - 1
        if job is None:
            return (None,) * (n + 1)
        convs, acc, ev, _keep = job
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        need = ctx.needs_input_grad
        return (None, *_unpack_param_grads(convs, acc, lambda j: need[1 + j]))


def _side_stream(device):
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _convs_of(ub):
    """The update block's packed convolutions in the order their parameters enter UpdateLoopFn."""
    e, g = ub.encoder, ub.gru
    return [e._c1p, e._c2, e._f1, e._f2, e._cv, g._zr_hm[0], g._q_hm[0], g._zr_hm[1], g._q_hm[1], ub._heads, ub._flow2, ub._mask2]


def loop_params(ub) -> List[Optional[Tensor]]:
    out = []
    for pc in _convs_of(ub):
        out += pc.params()
    return out


def eligible(ub, corr_fn, net, gru_pre) -> bool:
    """The fused node covers the default configuration: split conv formats (their gradient kernels return the bias gradient
    and scale by max|g|), the context share of the gates computed once, 128-channel state."""
    return (ENABLED and gru_pre is not None and ops.w_format() in (_hip.W_F16X3, _hip.W_F16) and net.shape[3] == 128
            and net.is_cuda and corr_fn.pyr is not None)


def _fwd(pc, xs, out, act=ACT_NONE, res=None, out_scale=1.0):
    w, b = pc.get()
    if not isinstance(xs, (list, tuple)):
        xs = [xs]
    return ops.conv2d(xs, w, b, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, act=ACT_NONE if res is not None else act, out=out, res=res,
                      act_res=act if res is not None else ACT_NONE, out_scale=out_scale, w_fmt=pc.fmt, dilation=pc.dil,
                      w_frag=pc.frag() if pc.dma_f32_ok(xs) else None)


def _dgrad(pc, g, amax, cin_tot, out, res=None):
    """Input gradient of `pc` over its (concatenated) input channels: a forward convolution over g with the flipped rows."""
    wd, dfmt = pc.get_dgrad()
    d = pc.dil
    return ops.conv2d([g], wd, None, cin_tot, pc.kh, pc.kw, 1, (d * (pc.kh - 1) - pc.pad[0], d * (pc.kw - 1) - pc.pad[1]),
                      w_fmt=dfmt, x_amax=amax if dfmt else None, dilation=d, out=out, res=res, w_frag=pc.frag_dgrad())


def _forward_fused(ub, corr_fn, coords1, T, net0, pre):
    """The recorded forward on the INFERENCE kernels (raft.py:218-231 through update_block.BasicUpdateBlock.run's split-pair route):
    eleven launches per iteration instead of nineteen.  What the backward needs is kept as it is produced - convolution inputs in
    the split-pair format they travel in (the backward turns the stacks back into fp32 with one pass each: (x0 + x1) / 4, 22
    significant bits), the GRU's gates z, r, q in fp32 from the recorded form of the pass kernel, the heads' hidden tensor in
    fp32; r * h and the 576-channel up-sampling mask are never stored (the backward recomputes them for all iterations at once)."""
    b, h, w, _ = net0.shape
    dev = net0.device
    enc, gru = ub.encoder, ub.gru

    def stack(c, n=T, zero=False):
        return (torch.zeros if zero else torch.empty)((n * b, h, w, c), dtype=torch.float32, device=dev)

    S = dict(fused=True, coords=torch.empty((T, b, h, w, 2), dtype=torch.float32, device=dev),
             corr=stack(352, zero=True), c1=stack(256), c2f2=stack(256), f1=stack(128), flow4=stack(4, T + 1), motion=stack(128),
             hin=stack(128, T + 1), hmid=stack(128), z=[stack(128), stack(128)], r=[stack(128), stack(128)], q=[stack(128), stack(128)], hid=stack(512))
    S["hin"][:b].copy_(net0)
    hcur = S["hin"][:b]
    hs = ops.split_copy(hcur)
    ops.coords_step(coords1, None, S["flow4"][:b], None)                       # flow = coords1 - coords0 (raft.py:219)
    wf = [(zc.frag(), qc.frag(), zc.get()[1], qc.get()[1], zc.fmt) for zc, qc in zip(gru._zr_hm, gru._q_hm)]
    outs = []
    for t in range(T):
        lo, hi = t * b, (t + 1) * b
        S["coords"][t].copy_(coords1)
        corr = S["corr"][lo:hi]
        ops.corr_lookup_tiled(corr_fn.pyr, S["coords"][t], out=corr[..., :324])
        c2f2, motion = S["c2f2"][lo:hi], S["motion"][lo:hi]
        # motion encoder (update.py:89-97): every tensor between its convolutions leaves as a split pair
        c1 = enc._c1p(corr, act=ACT_RELU, y_split=True, out=S["c1"][lo:hi])
        cor = enc._c2(c1, act=ACT_RELU, y_split=True, out=c2f2[..., :192])
        f1 = enc._f1(S["flow4"][lo:hi], act=ACT_RELU, y_split=True, out=S["f1"][lo:hi])
        flo = enc._f2(f1, act=ACT_RELU, y_split=True, out=c2f2[..., 192:])
        enc._cv([cor, flo], act=ACT_RELU, out=motion[..., :126], y_split=True, ep_motion_tail=coords1)      # + torch.cat([out, flow]) (update.py:97)
        mos = ops.SplitT(motion)
        # SepConvGRU (update.py:45-60): one launch per pass; z, r, q stay for the backward
        for k in range(2):
            hnew = S["hmid"][lo:hi] if k == 0 else S["hin"][hi:hi + b]
            wzr, wq, bzr, bq, fmt = wf[k]
            hcur, hs = ops.gru_pass(k, hs, mos, hcur, pre[k][0], pre[k][1], wzr, wq, bzr, bq, fmt, y=hnew,
                                    gates=(S["z"][k][lo:hi], S["r"][k][lo:hi], S["q"][k][lo:hi]))
        # heads (update.py:121-135): the flow head's last convolution takes the coordinate step, the mask head's the up-sampling
        hid = ub._heads(hs, act=ACT_RELU, out=S["hid"][lo:hi])
        ub._flow2(hid[..., :256], ep_coords=(coords1, S["flow4"][hi:hi + b]))
        outs.append(ub.upsample(hid[..., 256:], S["flow4"][hi:hi + b]))
    return S, outs


def _materialise(ub, S, T, b):
    """The stacks the backward reads, as fp32 tensors: after the conv-by-conv forward they are there already (plus r * h); after
    the fused forward the split-pair stacks are turned back (one pass each), r * h = r (.) h and the up-sampling mask
    0.25 * conv1x1(hidden) are recomputed for ALL iterations at once."""
    if not S.get("fused"):
        return S
    S = dict(S)
    for k in ("c1", "c2f2", "f1", "motion"):
        S[k] = ops.split_copy(S[k], to_split=False)
    hprev = [S["hin"][:T * b], S["hmid"]]
    S["rh"] = [ops.gru_rh(S["r"][k], hprev[k]) for k in range(2)]
    up = torch.empty(S["hid"].shape[:3] + (576,), dtype=torch.float32, device=S["hid"].device)
    _fwd(ub._mask2, S["hid"][..., 256:], up, out_scale=0.25)
    S["upmask"] = up
    S["fused"] = False
    return S


class UpdateLoopFn(torch.autograd.Function):
    """(net0, zr_pre1, q_pre1, zr_pre2, q_pre2, fmap1, fmap2, *params) -> the T up-sampled flows.  coords1 is advanced in
    place (never differentiated, raft.py:216/220).  With a LoopParamGate, `params` is its token alone and `box` the gate's
    mailbox: the parameter gradients then leave through the gate (see DEFER_PARAM_GRADS)."""

    @staticmethod
    def forward(ctx, ub, corr_fn, coords1, iters, box, net0, zr1, q1, zr2, q2, fmap1, fmap2, *params):
        b, h, w, _ = net0.shape
        T, dev = iters, net0.device
        ctx.ub, ctx.corr_fn, ctx.T, ctx.geom = ub, corr_fn, T, (b, h, w)
        ctx.box = box
        ctx.param_need = [q is not None and q.requires_grad for q in loop_params(ub)] if box is not None else None
        ctx.pre_shapes = [tuple(t.shape) for t in (zr1, q1, zr2, q2)]
        ctx.save_for_backward(fmap1, fmap2)
        ctx.set_materialize_grads(False)
        if FUSED_FWD and ops.w_format() == _hip.W_F16X3:
            ctx.S, outs = _forward_fused(ub, corr_fn, coords1, T, net0, [(zr1, q1), (zr2, q2)])
            return tuple(outs)

        def stack(c, n=T, zero=False):
            return (torch.zeros if zero else torch.empty)((n * b, h, w, c), dtype=torch.float32, device=dev)

        S = dict(coords=torch.empty((T, b, h, w, 2), dtype=torch.float32, device=dev),
                 corr=stack(352, zero=True), c1=stack(256), c2f2=stack(256), f1=stack(128), flow4=stack(4, T + 1), motion=stack(128),
                 hin=stack(128, T + 1), hmid=stack(128), zr=[stack(256), stack(256)], rh=[stack(128), stack(128)],
                 q=[stack(128), stack(128)], hid=stack(512), upmask=stack(576))
        enc, gru = ub.encoder, ub.gru
        pre = [(zr1, q1), (zr2, q2)]
        S["hin"][:b].copy_(net0)
        ops.coords_step(coords1, None, S["flow4"][:b], None)                       # flow = coords1 - coords0 (raft.py:219)
        outs = []
        for t in range(T):
            lo, hi = t * b, (t + 1) * b
            S["coords"][t].copy_(coords1)            # coords1 moves on in place: the backward scatter needs this iteration's
            corr = S["corr"][lo:hi]
            ops.corr_lookup_tiled(corr_fn.pyr, S["coords"][t], out=corr[..., :324])
            c2f2, motion = S["c2f2"][lo:hi], S["motion"][lo:hi]
            # motion encoder (update.py:89-97)
            _fwd(enc._c1p, corr, S["c1"][lo:hi], ACT_RELU)
            _fwd(enc._c2, S["c1"][lo:hi], c2f2[..., :192], ACT_RELU)
            _fwd(enc._f1, S["flow4"][lo:hi], S["f1"][lo:hi], ACT_RELU)
            _fwd(enc._f2, S["f1"][lo:hi], c2f2[..., 192:], ACT_RELU)
            _fwd(enc._cv, c2f2, motion[..., :126], ACT_RELU)
            ops.coords_step(coords1, None, None, motion[..., 126:])              # torch.cat([out, flow]) (update.py:97)
            # SepConvGRU (update.py:45-60) over [h, motion]; the context share of the gates is the pre-activation addend
            hcur = S["hin"][lo:hi]
            for k in range(2):
                zr, rh, q = S["zr"][k][lo:hi], S["rh"][k][lo:hi], S["q"][k][lo:hi]
                hnew = S["hmid"][lo:hi] if k == 0 else S["hin"][hi:hi + b]
                _fwd(gru._zr_hm[k], [hcur, motion], zr, ACT_SIGMOID, res=pre[k][0])
                ops.gru_rh(zr[..., 128:], hcur, out=rh)
                _fwd(gru._q_hm[k], [rh, motion], q, ACT_TANH, res=pre[k][1])
                ops.gru_blend(zr[..., :128], q, hcur, out=hnew)
                hcur = hnew
            # heads (update.py:121-135)
            hid = S["hid"][lo:hi]
            _fwd(ub._heads, hcur, hid, ACT_RELU)
            delta = _fwd(ub._flow2, hid[..., :256], None)
            _fwd(ub._mask2, hid[..., 256:], S["upmask"][lo:hi], out_scale=0.25)
            ops.coords_step(coords1, delta, S["flow4"][hi:hi + b], None)         # coords1 += delta; the new flow (raft.py:223)
            outs.append(ops.upsample_flow(S["flow4"][hi:hi + b], S["upmask"][lo:hi]))
        zrs = S.pop("zr")
        S["z"], S["r"] = [zr[..., :128] for zr in zrs], [zr[..., 128:] for zr in zrs]
        ctx.S = S
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        ub, T = ctx.ub, ctx.T
        b, h, w = ctx.geom
        S = _materialise(ub, ctx.S, T, b)
        fmap1, fmap2 = ctx.saved_tensors
        enc, gru = ub.encoder, ub.gru
        dev = S["hin"].device
        npix = b * h * w
        need = ctx.needs_input_grad
        NF = 5                                           # non-tensor arguments of forward
        convs = _convs_of(ub)
        gated = ctx.box is not None
        pneed = ctx.param_need if gated else list(need[NF + 7:])
        # which convolutions want parameter gradients (freeze_self: the flow head alone may train)
        pos, want_w = 0, []
        for pc in convs:
            n = len(pc.params())
            want_w.append(any(pneed[pos + i] for i in range(n)))
            pos += n
        partial = any(d is None for d in douts)
        mk = torch.zeros if partial else torch.empty

        def gstack(c, zero=False):
            return (torch.zeros if zero else mk)((T * b, h, w, c), dtype=torch.float32, device=dev)

        G = dict(mask=gstack(576), dflow=gstack(4, True), hid=gstack(512), zr=[gstack(256), gstack(256)], q=[gstack(128), gstack(128)],
                 m=gstack(128), c2f2=gstack(256), c1=gstack(256), f1=gstack(128), dcorr=gstack(352))
        NW = 12
        words = torch.zeros((T, NW), dtype=torch.int32, device=dev)
        W_MASK, W_FLOW, W_HID, W_Q2, W_ZR2, W_Q1, W_ZR1, W_M, W_C2, W_F2, W_C1, W_F1 = range(NW)
        dh = torch.empty((b, h, w, 128), dtype=torch.float32, device=dev)
        dm = torch.empty((b, h, w, 128), dtype=torch.float32, device=dev)
        dhid = torch.empty((b, h, w, 512), dtype=torch.float32, device=dev)
        dqc = torch.empty((b, h, w, 256), dtype=torch.float32, device=dev)
        dzc = torch.empty((b, h, w, 256), dtype=torch.float32, device=dev)
        d_c2f2 = torch.empty((b, h, w, 256), dtype=torch.float32, device=dev)
        d_c1 = torch.empty((b, h, w, 256), dtype=torch.float32, device=dev)
        d_f1 = torch.empty((b, h, w, 128), dtype=torch.float32, device=dev)
        dh_valid = False
        st = ops._stream
        p = ops._p
        # ---- weight gradients: per convolution ONE buffer for all iterations, filled by a few launches over slabs of WCHUNK
        # iterations (the stacks are ordinary batches of images) that run on a SIDE STREAM beside the input-gradient chain of the
        # earlier iterations: that chain is eleven dependent launches of ~200 blocks per iteration, the slab contractions are what
        # fills the rest of the chip meanwhile.  The kernels accumulate with atomics, so the slabs simply add up.
        hin, hmid, mot = S["hin"][:T * b], S["hmid"], S["motion"]
        jobs = [(S["corr"], G["c1"], W_C1), (S["c1"], G["c2f2"][..., :192], W_C2), (S["flow4"][:T * b], G["f1"], W_F1),
                (S["f1"], G["c2f2"][..., 192:], W_F2), (S["c2f2"], G["m"], W_M),
                ([hin, mot], G["zr"][0], W_ZR1), ([S["rh"][0], mot], G["q"][0], W_Q1), ([hmid, mot], G["zr"][1], W_ZR2),
                ([S["rh"][1], mot], G["q"][1], W_Q2), (S["hin"][b:], G["hid"], W_HID), (S["hid"][..., :256], G["dflow"], W_FLOW),
                (S["hid"][..., 256:], G["mask"], W_MASK)]
        acc = []
        for pc, (xs, g, wi), want in zip(convs, jobs, want_w):
            if not want:
                acc.append(None)
                continue
            kdim = pc.kh * pc.kw * sum(x.shape[3] for x in (xs if isinstance(xs, list) else [xs]))
            z = torch.zeros(pc.cout * kdim + pc.cout, dtype=torch.float32, device=dev)
            acc.append((z[:pc.cout * kdim].view(pc.cout, kdim), z[pc.cout * kdim:]))
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if WGRAD_SIDE_STREAM else main
        pending_hi = T                      # iterations [t, pending_hi) have complete gradient stacks and no weight-gradient launch yet

        def wgrad_slab(lo_t, hi_t):
            """Weight gradients of iterations lo_t .. hi_t - 1 (their G slices are complete on the main stream)."""
            if hi_t <= lo_t:
                return
            if side is not main:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
            with torch.cuda.stream(side):
                wmax = words[lo_t:hi_t].amax(0, keepdim=True).contiguous() if hi_t - lo_t > 1 else words[lo_t:hi_t]
                for pc, (xs, g, wi), a_ in zip(convs, jobs, acc):
                    if a_ is None:
                        continue
                    xs = xs if isinstance(xs, list) else [xs]
                    ops.conv2d_wgrad([x[lo_t * b:hi_t * b] for x in xs], g[lo_t * b:hi_t * b], pc.cout, pc.kh, pc.kw, pc.stride, pc.pad,
                                     g_amax=wmax[0, wi:wi + 1], want_db=True, dw=a_[0], db=a_[1], dilation=pc.dil)

        for t in range(T - 1, -1, -1):
            lo, hi = t * b, (t + 1) * b
            wd = lambda i, t=t: words[t, i:i + 1]                 # noqa: E731
            dout = douts[t]
            if dout is not None:
                # convex up-sampling (raft.py:159-170) -> d delta (4-channel, zero padded), .25 * d mask
                _hip.call("ff_upsample_flow_bwd_ex", p(dout.contiguous()), p(S["flow4"][hi:hi + b]), 4, p(S["upmask"][lo:hi]), 576,
                          p(G["dflow"][lo:hi]), 4, p(G["mask"][lo:hi]), 0.25, p(wd(W_MASK)), b, h, w, st())
                _hip.call("ff_act_bwd", p(G["dflow"][lo:hi]), 4, None, 0, p(G["dflow"][lo:hi]), 4, npix, 4, 4, ACT_NONE, 1.0, p(wd(W_FLOW)), st())
                _dgrad(ub._mask2, G["mask"][lo:hi], wd(W_MASK), 256, dhid[..., 256:])
                _dgrad(ub._flow2, G["dflow"][lo:hi], wd(W_FLOW), 256, dhid[..., :256])
                ops.act_bwd_into(dhid, S["hid"][lo:hi], ACT_RELU, G["hid"][lo:hi], wd(W_HID))
                _dgrad(ub._heads, G["hid"][lo:hi], wd(W_HID), 128, dh, res=dh if dh_valid else None)
                dh_valid = True
            if not dh_valid:
                continue                                     # nothing reaches this iteration (its G slices are zero)
            motion = S["motion"][lo:hi]
            later_zc = None
            for k in (1, 0):
                hprev = S["hin"][lo:hi] if k == 0 else S["hmid"][lo:hi]
                zg, rg, q = S["z"][k][lo:hi], S["r"][k][lo:hi], S["q"][k][lo:hi]
                gzr, gq = G["zr"][k][lo:hi], G["q"][k][lo:hi]
                wz, wq = (W_ZR2, W_Q2) if k == 1 else (W_ZR1, W_Q1)
                _hip.call("ff_gru_bwd_blend", p(dh), 128, p(later_zc), 256 if later_zc is not None else 0, p(dm) if later_zc is not None else None, 128,
                          p(zg), ops._ld(zg), p(q), 128, p(hprev), 128, p(gzr), 256, p(gq), 128, p(dh), 128, p(wd(wz)), p(wd(wq)), npix, 128, st())
                _dgrad(gru._q_hm[k], gq, wd(wq), 256, dqc)                      # -> [d (r h) | d motion]
                _hip.call("ff_gru_bwd_rh", p(dqc), 256, p(rg), ops._ld(rg), p(hprev), 128, p(gzr[..., 128:]), 256, p(dh), 128, p(dm), 128,
                          1 if k == 1 else 0, p(wd(wz)), npix, 128, st())
                _dgrad(gru._zr_hm[k], gzr, wd(wz), 256, dzc)                    # -> [d h | d motion]
                later_zc = dzc
            _hip.call("ff_gru_bwd_out", p(dh), 128, p(dzc), 256, p(dm), 128, p(motion), 128, p(dh), 128, p(G["m"][lo:hi]), 128, 126, p(wd(W_M)),
                      npix, 128, st())
            # motion encoder (update.py:89-97), backwards
            _dgrad(enc._cv, G["m"][lo:hi], wd(W_M), 256, d_c2f2)
            gc2f2 = G["c2f2"][lo:hi]
            ops.act_bwd_into(d_c2f2[..., :192], S["c2f2"][lo:hi][..., :192], ACT_RELU, gc2f2[..., :192], wd(W_C2))
            ops.act_bwd_into(d_c2f2[..., 192:], S["c2f2"][lo:hi][..., 192:], ACT_RELU, gc2f2[..., 192:], wd(W_F2))
            _dgrad(enc._c2, gc2f2[..., :192], wd(W_C2), 256, d_c1)
            ops.act_bwd_into(d_c1, S["c1"][lo:hi], ACT_RELU, G["c1"][lo:hi], wd(W_C1))
            _dgrad(enc._c1p, G["c1"][lo:hi], wd(W_C1), 352, G["dcorr"][lo:hi])
            if want_w[2]:        # convf1's input is the flow (no gradient): only its weights need d f1
                _dgrad(enc._f2, gc2f2[..., 192:], wd(W_F2), 128, d_f1)
                ops.act_bwd_into(d_f1, S["f1"][lo:hi], ACT_RELU, G["f1"][lo:hi], wd(W_F1))
            if pending_hi - t >= WCHUNK or t == 0:
                wgrad_slab(t, pending_hi)
                pending_hi = t
        # ---- everything below runs once per pass ----
        grads: List[Optional[Tensor]] = [None] * NF
        grads.append(dh if (need[NF] and dh_valid) else None)                      # d net0
        for i, (key, k) in enumerate((("zr", 0), ("q", 0), ("zr", 1), ("q", 1))):   # the context share of the gates: sum over t
            if need[NF + 1 + i] and dh_valid:
                src = G[key][k]
                dst = torch.empty(ctx.pre_shapes[i], dtype=torch.float32, device=dev)
                _hip.call("ff_sum_stack", p(src), T, src.numel() // T, p(dst), st())
                grads.append(dst)
            else:
                grads.append(None)
        # lookup scatter + pooling chain + the two volume contractions (corr.py:29-60 backward)
        if (need[NF + 5] or need[NF + 6]) and dh_valid:
            pyr = ctx.corr_fn.pyr
            cl, dl = [S["coords"][t] for t in range(T)], [G["dcorr"][t * b:(t + 1) * b] for t in range(T)]
            d0 = None
            if T <= ops.LOOKUP_BWD_ALL_MAX and ops.lookup_bwd_all_fits(pyr.h0, pyr.w0) and fn._LOOKUP_BWD_ALL:
                d0 = ops.corr_lookup_tiled_bwd_all(cl, dl, pyr.h0, pyr.w0)
            if d0 is None:
                gp = ops.TiledPyramid.empty(pyr.levels[0].shape[0], pyr.h0, pyr.w0, False, dev, zero=True)
                for c, d in zip(cl, dl):
                    ops.corr_lookup_tiled_bwd(gp, c, d)
                ops.corr_pyramid_tiled_bwd(gp)
                d0 = gp.levels[0]
            df1, df2 = ops.corr_volume_bwd(d0.view(b, h * w, -1), fmap1, fmap2, tiled=True)
            grads += [df1 if need[NF + 5] else None, df2 if need[NF + 6] else None]
        else:
            grads += [None, None]
        # parameter gradients: the slabs' sums, un-packed per convolution group.  Through the gate: left in its mailbox with the
        # event behind the last slab (the token's own gradient stays None); otherwise the side stream joins here
        if not dh_valid:
            acc = [None] * len(acc)
        if gated:
            ev = None
            if side is not main:
                ev = torch.cuda.Event()
                ev.record(side)
            # (the stacks the slabs read were allocated on the main stream: they stay referenced until the gate has waited for the
            # event, or the allocator would hand their memory to the encoders' backward while the side stream still reads it)
            ctx.box["job"] = (convs, acc, ev, (S, G, words))
            return tuple(grads + [None])
        if side is not main:
            main.wait_stream(side)
        return tuple(grads + _unpack_param_grads(convs, acc, lambda j: pneed[j]))
