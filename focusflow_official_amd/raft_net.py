"""RAFT graph on the HIP path (raft.py:40-236, the 'parallel' inside-fusion build)."""
import os

import torch
import torch.nn as nn

from . import _hip, cce, fn, ops, train_loop
from .cce import BasicParallelFusionLayer, train_streams
from .corr_block import CorrBlock
from .ops import ACT_RELU, ACT_TANH
from .update_block import BasicUpdateBlock, split_activations

_COORDS_EPILOGUE = True   # coords1 += delta inside the flow head's last convolution (inference)
_MASK_UPSAMPLE = True     # mask conv 2 + convex up-sampling as one kernel (inference)
_GRU_CTX_ONCE = True      # the context features' share of the GRU gates once per forward (SepConvGRU.prepare)
_STREAMS_MIN_PIXELS = int(os.environ.get("FF_STREAMS_MIN_PIXELS", "700000"))   # (a test forces the streams on at its small sizes)   # below: host-bound, the fork / join events cost more than they gain
_ENC_STREAMS = True       # cnet on a second stream beside fnet


class RAFT(nn.Module):
    def __init__(self, in_channels=3, small=False, dropout=0., alternate_corr=False, abandon_fnet=False,
                 inside_fusion=None, fuse_cnet=False, cfg=None):
        super().__init__()
        if small or abandon_fnet or inside_fusion != "parallel" or not fuse_cnet:
            raise NotImplementedError(
                "the HIP path builds the configuration every shipped FF-RAFT experiment uses: "
                "small=False, inside_fusion='parallel', fuse_cnet=True (SURVEY §2.1)")
        self.small, self.abandon_fnet, self.inside_fusion, self.fuse_cnet, self.cfg = small, abandon_fnet, inside_fusion, fuse_cnet, cfg
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        self.corr_levels, self.corr_radius = 4, 4
        self.dropout = dropout
        # ALT_CORR selects an on-the-fly correlation in the reference (its alt_cuda_corr extension is not vendored
        # there, corr.py:5-9); here the materialised pyramid is always used - say so, the caller may have set the flag to
        # bound memory (B * Q^2 * 5.3 bytes in fp32, half that with corr_pyramid_dtype = "fp16")
        self.alternate_corr = alternate_corr
        if alternate_corr:
            import warnings
            warnings.warn("alternate_corr=True: the HIP path has no on-the-fly correlation; the materialised all-pairs "
                          "pyramid is used (O((H*W/64)^2) memory per pair; corr_pyramid_dtype='fp16' halves it)")
        # storage type of the correlation pyramid: None = $FF_CORR_PYRAMID or "fp32"; "fp16" = BASELINE configs[4]
        self.corr_pyramid_dtype = None
        mc = cfg.TRAIN.MASK_CHANNEL
        self.fnet = BasicParallelFusionLayer(3, mc, output_dim=256, norm_fn="instance", dropout=dropout, cfg=cfg)
        self.cnet = BasicParallelFusionLayer(3, mc, output_dim=hdim + cdim, norm_fn="batch", dropout=dropout, cfg=cfg)
        self.update_block = BasicUpdateBlock(self.corr_levels, self.corr_radius, hidden_dim=hdim)
        # the always-on range guard's memory of this model (ops.guard_*): running max|x| of the two guarded encoder outputs, and
        # the sticky repair - the context features' convolutions on the exact-fp32 route
        self._guard_hist, self._guard_level, self._exact_ctx, self._guard_handled = False, 0.0, False, False

    # -- reference API (raft.py:104-148) ------------------------------------
    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def freeze_self(self, mode):
        if mode == "parallel":
            self.fnet.freeze_self(mode)
            self.cnet.freeze_self(mode)
            self.update_block.freeze_self(mode)

    def load_model(self, model_path, flag="all", strict=True):
        model_dict = {k.replace("module.", ""): v for k, v in torch.load(model_path).items()}
        if flag == "backend":
            for k in ("fnet.conv1.weight", "fnet.conv1.bias", "cnet.conv1.weight", "cnet.conv1.bias"):
                model_dict.pop(k)
            strict = False
        self.load_state_dict(model_dict, strict=strict)
        if flag == "all" and self.cfg.MODEL.LOAD_MODULE_TO_BRANCH:
            self.fnet.copy_to_branch()
            self.cnet.copy_to_branch()

    # -- forward ------------------------------------------------------------
    def forward(self, image1, image2, mask1=None, mask2=None, iters=12, flow_init=None, upsample=True,
                test_mode=False):
        """Inputs are NHWC4 tensors from ops.prep_input (already scaled to [-1,1]).
        Returns the reference's outputs in NCHW: a list of `iters` (B,2,H,W) flows,
        or (flow_low, flow_up) in test_mode.  raft.py:173-236."""
        b, hh, ww, _ = image1.shape
        h8, w8 = hh // 8, ww // 8
        ops.begin_forward(image1.device)     # one zeroed arena for this pass's norm statistics
        if torch.is_grad_enabled():
            fn.begin_graph(image1.device)   # one weight-gradient buffer per conv for this recorded pass (fn.GraphScope)
            cce.prepack(self, image1.device)   # the optimiser changed every parameter: all kernel layouts in one launch
        try:
            return self._forward(image1, image2, mask1, mask2, iters, flow_init, test_mode, b, hh, ww, h8, w8)
        finally:
            fn.end_graph()

    def _forward(self, image1, image2, mask1, mask2, iters, flow_init, test_mode, b, hh, ww, h8, w8):
        # both frames through fnet as ONE batch of 2B (InstanceNorm is per sample, so this is the same arithmetic as
        # the reference's two calls, raft.py:187-189): grids twice as large at the 1/4- and 1/8-resolution layers
        # (fewer ragged last waves of blocks) and half the launches
        # Inference: the context encoder (BatchNorm folded: convolutions only) on a second stream beside the feature
        # encoder, whose InstanceNorm statistics / apply passes are memory-bound - the two use different parts of the chip.
        # recorded passes: the whole update loop is one autograd node (train_loop.UpdateLoopFn); its parameters' gradients leave
        # through a gate node created HERE, before the encoders, so that its backward runs after theirs (train_loop.LoopParamGate)
        fused_train = (torch.is_grad_enabled() and train_loop.ENABLED and _GRU_CTX_ONCE and ops.w_format() in (_hip.W_F16X3, _hip.W_F16)
                       and not test_mode)
        loop_gate = None
        if fused_train and train_loop.defer_param_grads():
            lp = train_loop.loop_params(self.update_block)
            if all(q is None or q.is_cuda for q in lp) and any(q is not None and q.requires_grad for q in lp):
                box = {}
                loop_gate = (box, train_loop.LoopParamGate.apply(box, *lp))
        ops.policy.encoder_streams_ok = b * hh * ww >= _STREAMS_MIN_PIXELS
        two_streams = (_ENC_STREAMS and ops.policy.encoder_streams_ok and not ops.policy.single_stream and (not torch.is_grad_enabled() or train_streams())
                       )      # (also while a hipGraph is being captured: one level of forks is capturable; the branch forks inside each encoder are not - cce._branches)
        if two_streams:
            main = torch.cuda.current_stream()
            if getattr(self, "_enc_stream", None) is None:
                self._enc_stream = torch.cuda.Stream(device=image1.device)
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(self._enc_stream):
                self._enc_stream.wait_event(fork)
                cnet = self.cnet(image1, mask1)
                join = torch.cuda.Event()
                join.record(self._enc_stream)
            for t in (image1, mask1):
                t.record_stream(self._enc_stream)
        f12 = self.fnet(ops.cat_batch(image1, image2), ops.cat_batch(mask1, mask2))
        fmap1, fmap2 = f12[:b], f12[b:]
        self.fmap = fmap1
        # the fused update-loop node takes the feature maps themselves - the pyramid is then built outside the tape
        if fused_train:
            corr_fn = CorrBlock(fmap1.detach(), fmap2.detach(), radius=self.corr_radius, pyramid_dtype=self.corr_pyramid_dtype)
        else:
            corr_fn = CorrBlock(fmap1, fmap2, radius=self.corr_radius, pyramid_dtype=self.corr_pyramid_dtype)
        if two_streams:
            main.wait_event(join)
            cnet.record_stream(main)
        else:
            cnet = self.cnet(image1, mask1)
        ops.guard_probe(cnet, 0)        # the always-on range guard: the two encoder outputs (ops.guard_begin)
        ops.guard_probe(f12, 1)
        if ops.guard_careful(self):     # first forward on these weights, or the range is within a factor 4 of the limit: look NOW
            m_ctx, m_fmap = ops.guard_read_now()
            ops.guard_note(self, m_ctx, m_fmap)
            self._guard_handled = True
            if not (m_fmap < ops.X_LIMIT) or not (m_ctx < float("inf")):
                raise _hip.FocusFlowHipError(
                    f"RAFT.forward: an encoder output reached |x| = {max(m_ctx, m_fmap):.6g}; the fp16-split conv formats need |x| < {ops.X_LIMIT:g} "
                    "(csrc/ff_common.h).  Feature maps beyond it have no local exact route (the correlation values they produce overflow the "
                    "next layer as well), and an infinite context output means a layer INSIDE the encoder overflowed.  Run this checkpoint / "
                    "input with FF_CONV_PRECISION=fp32.")
            if not (m_ctx < ops.X_LIMIT) and not self._exact_ctx:
                if fn.recording(cnet):
                    raise _hip.FocusFlowHipError(
                        f"RAFT.forward: the context encoder's output reached |x| = {m_ctx:.6g} (limit {ops.X_LIMIT:g}) in a RECORDED pass; the "
                        "exact-fp32 repair covers inference only.  Train this checkpoint with FF_CONV_PRECISION=fp32.")
                import warnings
                warnings.warn(f"FF-RAFT: the context encoder's output reached |x| = {m_ctx:.6g} (limit of the fp16-split conv formats: "
                              f"{ops.X_LIMIT:g}); the convolutions that read it run on the exact-fp32 MFMA route from now on")
                self._exact_ctx = True
        taped = fn.recording(cnet)
        if taped:
            net = fn.ActFn.apply(cnet[..., :128], ACT_TANH)
            inp = fn.ActFn.apply(cnet[..., 128:], ACT_RELU)
        else:
            net = ops.empty_nhwc(b, h8, w8, 128, cnet)
            inp = ops.empty_nhwc(b, h8, w8, 128, cnet)
            ops.act_copy(cnet[..., :128], net, ACT_TANH)
            ops.act_copy(cnet[..., 128:], inp, ACT_RELU)
        coords1 = ops.coords_init(b, h8, w8, cnet, flow_init)      # never differentiated (raft.py:216)
        # the context features' share of the GRU gate convolutions does not change over the iterations
        gru_pre = self.update_block.gru.prepare(inp, exact=self._exact_ctx and not taped) if _GRU_CTX_ONCE and (taped or fused_train or not torch.is_grad_enabled()) else None
        if fused_train:
            lp = train_loop.loop_params(self.update_block)
            pre = [t for zq in gru_pre for t in zq]
            if fn.recording(net, *pre, fmap1, fmap2, *lp) and train_loop.eligible(self.update_block, corr_fn, net, gru_pre):
                if loop_gate is not None:
                    return list(train_loop.UpdateLoopFn.apply(self.update_block, corr_fn, coords1, iters, loop_gate[0], net, *pre,
                                                              fmap1.contiguous(), fmap2.contiguous(), loop_gate[1]))
                return list(train_loop.UpdateLoopFn.apply(self.update_block, corr_fn, coords1, iters, None, net, *pre, fmap1.contiguous(), fmap2.contiguous(), *lp))
            if fn.recording(fmap1, fmap2):      # (not eligible after all: the per-operation tape needs the pyramid on the tape)
                corr_fn = CorrBlock(fmap1, fmap2, radius=self.corr_radius, pyramid_dtype=self.corr_pyramid_dtype)
        flow4, flow_up, flow_predictions = self._loop(net, inp, corr_fn, coords1, gru_pre, iters, b, h8, w8, taped, test_mode)
        if test_mode:
            return ops.nhwc_to_nchw(flow4[..., :2]), flow_up
        return flow_predictions

    def _loop(self, net, inp, corr_fn, coords1, gru_pre, iters, b, h8, w8, taped, test_mode):
        """raft.py:218-236 for one batch (or batch slice): returns (flow4, flow_up, flow_predictions)."""
        out = {}
        for _ in self._loop_steps(net, inp, corr_fn, coords1, gru_pre, iters, b, h8, w8, taped, test_mode, out):
            pass
        return out["flow4"], out["flow_up"], out["preds"]

    def _loop_steps(self, net, inp, corr_fn, coords1, gru_pre, iters, b, h8, w8, taped, test_mode, out):
        """The same as a generator that yields after every iteration; results land in `out`.  (Two half batches on two streams,
        the iterations issued alternately, were +2.5-3 % under graph replay in rounds 3 / 4 - and halved the lookup's launches;
        removed in round 5, docs/history.md.)"""
        cnet = net
        flow_predictions = []
        flow_up = None
        flow4 = None
        # The reference up-samples every iteration and, in test_mode, throws 11 of the 12 results away
        # (raft.py:226-236).  Default: do the same work.  skip_unused_upsample (opt-in, inference only) computes
        # the mask head + convex up-sampling for the last iteration only; flow_low / flow_up are bit-identical.
        lazy = bool(getattr(self, "skip_unused_upsample", False)) and test_mode and not taped
        # inference: the mask head's 1x1 convolution and the convex up-sampling run as one kernel (no 576-channel mask tensor)
        fused_coords = _COORDS_EPILOGUE and not taped and not torch.is_grad_enabled() and coords1.is_contiguous()
        fused_up = _MASK_UPSAMPLE and not taped and not torch.is_grad_enabled() and ops.w_format() in (_hip.W_F16X3, _hip.W_F16)
        if split_activations() and gru_pre is not None and not taped and coords1.is_contiguous():
            # inference: split-pair activations between the update block's convolutions (update_block.split_activations);
            # the state travels as (fp32 for the element-wise steps, split-pair for the convolutions, coords1)
            net = (net, ops.split_copy(net), coords1)
        for it in range(iters):
            # coords1 is advanced in place by ff_coords_step (raw pointer: autograd cannot see it), so a
            # recorded lookup keeps its own snapshot for the backward scatter
            corr = corr_fn(coords1.clone() if taped else coords1)
            if it == 0 or taped:   # later iterations: the flow written with the coordinate update below is this very tensor
                flow4 = ops.empty_nhwc(b, h8, w8, 4, cnet)
                ops.coords_step(coords1, None, flow4, None)               # flow = coords1 - coords0
            fill = lambda motion, c=coords1: ops.coords_step(c, None, None, motion[..., 126:])  # noqa: E731
            need_mask = not lazy or it == iters - 1
            if fused_coords and need_mask:      # the flow head's last convolution takes the coordinate step with it
                nflow4 = ops.empty_nhwc(b, h8, w8, 4, cnet)
                net, up_mask, delta = self.update_block.run(net, inp, corr, flow4, fill, need_mask, gru_pre, defer_mask=fused_up, coords_out=(coords1, nflow4))
                flow4 = nflow4
            else:
                net, up_mask, delta = self.update_block.run(net, inp, corr, flow4, fill, need_mask, gru_pre, defer_mask=fused_up)
                flow4 = ops.empty_nhwc(b, h8, w8, 4, cnet)
                ops.coords_step(coords1, delta.detach(), flow4, None)         # coords1 += delta
            if need_mask:
                if fused_up:
                    # (issuing this behind the NEXT iteration's lookup - so that the lookup's predecessor is the flow head's
                    # 0.2 MB instead of these 25 MB of dirty output - was measured: 540 -> 532 pairs/s, lookup 19.4 -> 19.8 us)
                    flow_up = self.update_block.upsample(up_mask, flow4)
                elif fn.recording(delta, up_mask):
                    flow_up = fn.UpsampleFn.apply(flow4, delta, up_mask)
                else:
                    flow_up = ops.upsample_flow(flow4, up_mask)
                flow_predictions.append(flow_up)
            out.update(flow4=flow4, flow_up=flow_up, preds=flow_predictions)
            yield
