"""Tensor-level wrappers over the C ABI (device pointers in, device pointers out).

Activations are NHWC fp32 torch tensors of shape (B, H, W, C) whose last dim is
contiguous; a tensor may be a channel slice of a wider buffer (stride(2) = ld).
torch is used for allocation and stream handles only — all arithmetic happens
in libfocusflow_hip.so.
"""
import ctypes as C
import math
import os
import threading
from typing import List, Optional, Sequence

import torch

from . import _hip
from ._hip import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, FFConvParams  # noqa: F401

Tensor = torch.Tensor


# The current stream of the current device as a raw hipStream_t.  torch.cuda.current_stream() builds a Stream object through
# several Python layers (~8 us; a one-pair forward asks ~350 times and is host-bound): the two C entry points behind it
# are used directly where this torch has them.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Arithmetic of the forward convolutions / correlation volume:
#   "f16x3" (default) fp16-split operands, 3 f16 MFMAs: fp32-level accuracy (see csrc/conv_split.hip)
#   "fp32"  exact fp32 MFMA            "f16"  plain fp16 operands (reduced precision, throughput mode)
_PRECISIONS = {"fp32": _hip.W_F32, "f16x3": _hip.W_F16X3, "f16": _hip.W_F16}
_conv_precision = os.environ.get("FF_CONV_PRECISION", "f16x3")


def set_conv_precision(name: str):
    global _conv_precision
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    _conv_precision = name


def conv_precision() -> str:
    return _conv_precision


def w_format() -> int:
    return _PRECISIONS[_conv_precision]


def pack_split(rows_f32: Tensor) -> Tensor:
    """fp32 [rows][K] -> split fp16 rows (uint8 view [rows][ceil(K/32)*128])."""
    _require_gpu(rows_f32)
    rows, k = rows_f32.shape
    assert rows_f32.is_contiguous()
    dst = torch.empty((rows, (k + 31) // 32 * 128), dtype=torch.uint8, device=rows_f32.device)
    _hip.call("ff_pack_split_f16", _p(rows_f32), _p(dst), rows, k, _stream())
    return dst


def pack_frag16(w_split: Tensor, rows: int) -> Tensor:
    """Split rows (pack_split) -> MFMA-fragment order (ff_pack_frag16) for FFConvParams.w_frag."""
    nkc = w_split.shape[1] // 128
    dst = torch.empty(((rows + 15) // 16) * nkc * 2048, dtype=torch.uint8, device=w_split.device)
    _hip.call("ff_pack_frag16", _p(w_split), _p(dst), rows, nkc, _stream())
    return dst


class _StreamPolicy(threading.local):
    """Per-THREAD stream policy of the forward passes (two models driven from two threads do not see each other's).
    single_stream: set by a caller that brackets single launches with events (bench.py's roofline_conv leg) - every
    forward then runs on ONE stream, so that the bracketed durations add up to wall time instead of overlapping.
    encoder_streams_ok: set per forward by RAFT._forward - below ~3.5 pairs of 384x512 a step is host-bound and the
    fork / join events of the encoder streams cost more than the overlap gains (configs[4] at one pair: 16.4 -> 16.7 ms)."""
    single_stream = False
    encoder_streams_ok = True


policy = _StreamPolicy()

# Optional per-launch timing of ONE entry point with HIP events recorded on the
# launch stream (bench.py's roofline leg).  Off unless profile_begin() is called.
_prof_on = False
_prof_events = {}
_prof_notes = {}      # label -> per-launch annotation (conv: useful FLOP of the launch)


def profile_begin(*labels: str):
    """Time every launch carrying one of `labels` with a HIP event pair on the launch stream."""
    global _prof_on
    _prof_on = True
    _prof_events.clear()
    _prof_notes.clear()
    for lb in labels:
        _prof_events[lb] = []
        _prof_notes[lb] = []


def profile_end():
    """-> {label: [per-launch milliseconds]} (synchronises)."""
    global _prof_on
    _prof_on = False
    torch.cuda.synchronize()
    out = {lb: [a.elapsed_time(b) for a, b in ev] for lb, ev in _prof_events.items()}
    _prof_events.clear()
    return out


def profile_notes(label):
    """Annotations recorded next to the timings of `label` (read before the next profile_begin)."""
    return list(_prof_notes.get(label, []))


TIME_LOOKUP, TIME_CORR_BUILD, TIME_PROBE = 1, 2, 3          # include/focusflow_hip.h: FF_TIME_*


def probe_memory_kernel(src: Tensor, dst: Tensor, seg_bytes: int, blocks: int, trips: int, salt: int):
    """ff_probe_memory_kernel: the memory-only companion of the lookup (measurement aid) -> (bytes read, bytes written)."""
    if not (src.is_cuda and dst.is_cuda):
        raise _hip.FocusFlowHipError("ff_probe_memory_kernel: device buffers only")
    rd, wr = C.c_longlong(0), C.c_longlong(0)
    _hip.call("ff_probe_memory_kernel", _p(src), src.numel() * src.element_size(), _p(dst), dst.numel() * dst.element_size(), seg_bytes,
              blocks, trips, salt, C.byref(rd), C.byref(wr), _stream())
    return rd.value, wr.value


def launch_timing_begin(*which: int):
    """Kernel-timestamp timing of the library's lookup / corr-build launches (ff_launch_timing_begin): HIP events bound
    to the dispatch itself, on whatever stream the launch goes to."""
    for w in which:
        _hip.call("ff_launch_timing_begin", w)


def launch_timing_end(which: int):
    """-> (launches, total_us, min_us, max_us) since launch_timing_begin; waits for the launches."""
    n, tot, lo, hi = C.c_longlong(0), C.c_double(0), C.c_double(0), C.c_double(0)
    _hip.call("ff_launch_timing_end", which, C.byref(n), C.byref(tot), C.byref(lo), C.byref(hi))
    return n.value, tot.value, lo.value, hi.value


def _timed_call(label, name, *args, note=None):
    if _prof_on and label in _prof_events:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.call(name, *args)
        b.record()
        _prof_events[label].append((a, b))
        _prof_notes[label].append(note)
    else:
        _hip.call(name, *args)


def _require_gpu(t: Tensor):
    if not t.is_cuda:
        raise _hip.FocusFlowHipError(
            "the FF-RAFT hot path runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise _hip.FocusFlowHipError(f"fp32 tensors expected, got {t.dtype}")


def _ld(t: Tensor) -> int:
    """floats per pixel of an NHWC (B,H,W,C) view; checks it is pixel-dense."""
    _require_gpu(t)
    b, h, w, c = t.shape
    ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else max(t.stride(0), c))
    if t.stride(3) != 1 or (w > 1 and h > 1 and t.stride(1) != w * ld) or (b > 1 and t.stride(0) != h * w * ld):
        raise _hip.FocusFlowHipError(f"not an NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def _p(t: Optional[Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def empty_nhwc(b, h, w, c, like: Tensor) -> Tensor:
    return torch.empty((b, h, w, c), dtype=torch.float32, device=like.device)


class SplitT:
    """An NHWC activation in the split-pair format (FF_FMT_SPLIT, include/focusflow_hip.h): shape, strides and bytes of its
    fp32 form, but every 32-channel chunk of a pixel holds [x0: 32 fp16 | x1: 32 fp16] - what a convolution loader makes
    of the fp32 values, written by the producer so that the consumer's patch goes L2 -> LDS by LDS-DMA.  Only convolutions
    read it (ops.conv2d sets FFConvParams.x_fmt); `.t` is the raw storage and means nothing as floats."""
    __slots__ = ("t",)

    def __init__(self, t: Tensor):
        self.t = t

    @property
    def shape(self):
        return self.t.shape

    @property
    def device(self):
        return self.t.device

    def __getitem__(self, idx):
        r = self.t[idx]
        off = (r.data_ptr() - self.t.data_ptr()) // 4
        ld = _ld(self.t)
        if (off % ld) % 32 or r.shape[3] % 32:
            raise _hip.FocusFlowHipError("a split-pair tensor can only be sliced at multiples of 32 channels")
        return SplitT(r)

    def record_stream(self, s):
        self.t.record_stream(s)

    def float(self) -> Tensor:
        """Back to fp32 ((x0 + x1) / 4: 22 significant bits) - tests and debugging."""
        return split_copy(self.t, to_split=False)


def _raw(x):
    return x.t if isinstance(x, SplitT) else x


def split_copy(x: Tensor, act: int = ACT_NONE, to_split: bool = True, out: Optional[Tensor] = None):
    """fp32 NHWC -> SplitT (ff_split_copy; an activation on the way), or - to_split=False - the raw storage of a SplitT back
    to fp32."""
    b, h, w, c = x.shape
    if out is None:
        out = empty_nhwc(b, h, w, c, x)
    _hip.call("ff_split_copy", _p(x), _ld(x), _p(out), _ld(out), b * h * w, c, act, 1 if to_split else 0, _stream())
    return SplitT(out) if to_split else out


# ----------------------------------------------------------------------------
def pack_conv_weight(w_oihw: Tensor, dst: Tensor, cin_pad: int, cout_offset: int = 0):
    """OIHW parameter -> rows [cout_offset, cout_offset+Cout) of dst [rows][KH*KW*cin_pad]."""
    _require_gpu(w_oihw)
    co, ci, kh, kw = w_oihw.shape
    assert dst.is_contiguous() and dst.shape[1] == kh * kw * cin_pad and dst.shape[0] >= cout_offset + co
    _hip.call("ff_pack_conv_weight", _p(w_oihw.contiguous()), co, ci, kh, kw, _p(dst), cin_pad, cout_offset, _stream())


def conv2d(xs: Sequence[Tensor], wpack: Tensor, bias: Optional[Tensor], cout: int, kh: int, kw: int,
           stride: int = 1, pad=(0, 0), act: int = ACT_NONE, out: Optional[Tensor] = None,
           res: Optional[Tensor] = None, act_res: int = ACT_NONE, ch_scale: Optional[Tensor] = None,
           ch_shift: Optional[Tensor] = None, out_scale: float = 1.0, w_fmt: int = 0, dilation: int = 1,
           x_amax: Optional[Tensor] = None, in_scale: Optional[Tensor] = None, in_shift: Optional[Tensor] = None,
           in_act: int = ACT_NONE, res2: Optional[Tensor] = None, res_split: int = 0,
           ep_rh: Optional[Tensor] = None, ep_split: int = 0, ep_blend=None, want_stats: bool = False, ep_coords=None,
           y_split=False, y2_split: bool = False, ep_motion_tail: Optional[Tensor] = None, w_frag: Optional[Tensor] = None):
    """Convolution over the channel-concatenation of `xs` (see FFConvParams).  want_stats: -> (out, stats) with the
    per-sample {sum, sum of squares} table of the output (what norm_stats(out, True) returns): from the convolution's own
    epilogue where the kernel can (FFConvParams.stats_part), else from a norm_stats pass.  res2 / res_split: output channels
    >= res_split take their residual from `res2` (paired 1x1 fusion convs).  `wpack` is fp32
    [Cout][K] (w_fmt 0) or the split rows of pack_split (w_fmt 1/2).  x_amax: device word holding the bits of
    max|x| (act_bwd): the split formats then scale the input by a power of two so that gradients fit fp16."""
    if isinstance(pad, int):
        pad = (pad, pad)
    fmts = [1 if isinstance(x, SplitT) else 0 for x in xs]
    xs = [_raw(x) for x in xs]
    x0 = xs[0]
    b, h, w, _ = x0.shape
    ho = (h + 2 * pad[0] - dilation * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad[1] - dilation * (kw - 1) - 1) // stride + 1
    if out is None:
        if y_split is not False:      # a split-pair chunk is 32 channels: the buffer holds Cout rounded up to 32 (the x1 half of the last chunk)
            full = empty_nhwc(b, ho, wo, (cout + 31) // 32 * 32, x0)
            out = full if full.shape[3] == cout else full[..., :cout]
        else:
            out = empty_nhwc(b, ho, wo, (cout + 3) // 4 * 4, x0)[..., :cout] if cout % 4 else empty_nhwc(b, ho, wo, cout, x0)
    p = FFConvParams()
    cin = 0
    for i, x in enumerate(xs):
        assert x.shape[:3] == x0.shape[:3]
        p.x[i] = x.data_ptr()
        p.x_ld[i] = _ld(x)
        p.x_c[i] = x.shape[3]
        p.x_gstride[i] = 0
        p.x_fmt[i] = fmts[i]
        cin += x.shape[3]
    kdim = kh * kw * cin if w_fmt == 0 else (kh * kw * cin + 31) // 32 * 128
    assert wpack.is_contiguous() and wpack.shape[-1] == kdim and wpack.shape[0] >= cout, \
        f"packed weight {tuple(wpack.shape)} vs Cout {cout}, K {kh * kw * cin} (format {w_fmt})"
    p.groups, p.B, p.H, p.W = 1, b, h, w
    p.w, p.w_gstride = wpack.data_ptr(), 0
    if w_frag is not None:      # the same weights in fragment order (pack_frag16): read by conv_dma.hip only
        p.w_frag = w_frag.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.ch_scale = ch_scale.data_ptr() if ch_scale is not None else None
    p.ch_shift = ch_shift.data_ptr() if ch_shift is not None else None
    p.out_scale = out_scale
    p.res = res.data_ptr() if res is not None else None
    p.res_ld = _ld(res) if res is not None else 0
    if res2 is not None:
        p.res2, p.res2_ld, p.res_split = res2.data_ptr(), _ld(res2), res_split
    if ep_rh is not None:       # FF_EP_GRU_RH: output channels >= ep_split leave multiplied by ep_rh (the z|r conv writes [z | r*h])
        p.ep_mode, p.ep_split, p.ep_a, p.ep_a_ld = 1, ep_split, ep_rh.data_ptr(), _ld(ep_rh)
    elif ep_coords is not None:  # FF_EP_COORDS: (coords1, flow4) - the flow head also takes the coordinate step
        c1, f4 = ep_coords
        assert c1.is_contiguous() and c1.shape == (b, ho, wo, 2) and f4.is_contiguous() and f4.shape == (b, ho, wo, 4)
        p.ep_mode, p.ep_a, p.ep_b = 3, c1.data_ptr(), f4.data_ptr()
    elif ep_blend is not None:  # FF_EP_GRU_BLEND: (z, h) -> the output is (1 - z) h + z v (the q conv writes the new state)
        z, hprev = ep_blend
        p.ep_mode, p.ep_a, p.ep_a_ld, p.ep_b, p.ep_b_ld = 2, z.data_ptr(), _ld(z), hprev.data_ptr(), _ld(hprev)
    elif ep_motion_tail is not None:   # FF_EP_MOTION_TAIL: channels Cout, Cout + 1 of the padded output = coords1 - pixel grid
        assert ep_motion_tail.is_contiguous() and ep_motion_tail.shape == (b, ho, wo, 2)
        p.ep_mode, p.ep_a = 4, ep_motion_tail.data_ptr()
    p.y, p.y_ld, p.y_gstride = out.data_ptr(), _ld(out), 0
    if y_split is not False:
        p.y_fmt, p.y_fmt_from = 1, (0 if y_split is True else int(y_split))
    out2 = None
    if y2_split:
        out2 = empty_nhwc(b, ho, wo, (cout + 31) // 32 * 32, x0)
        p.y2, p.y2_ld = out2.data_ptr(), _ld(out2)
    p.Ho, p.Wo, p.Cout = ho, wo, cout
    p.KH, p.KW, p.stride, p.pad_h, p.pad_w = kh, kw, stride, pad[0], pad[1]
    p.act, p.act_res, p.w_format = act, act_res, w_fmt
    p.dil_h = p.dil_w = dilation
    p.x_amax = x_amax.data_ptr() if x_amax is not None else None
    if in_scale is not None:   # norm_coeffs tables [B][Cin]: the input is read as in_act(x * scale + shift) (normalise-on-load)
        assert in_scale.shape == in_shift.shape == (b, cin) and in_scale.is_contiguous() and in_shift.is_contiguous()
        p.in_scale, p.in_shift, p.in_act = in_scale.data_ptr(), in_shift.data_ptr(), in_act
    assert out.shape[:3] == (b, ho, wo) and out.shape[3] >= cout
    if _range_word is not None and w_fmt and x_amax is None and in_scale is None:
        for x, f in zip(xs, fmts):
            if not f:          # (a split-pair tensor was range-checked as the fp32 values its producer held)
                _range_probe(x)
    klen = cin * kh * kw
    plain = not any(fmts) and y_split is False and not y2_split                # (the split-pair kernels have no K splits)
    if plain and w_fmt and not p.ep_mode and b * ho * wo <= 16384 and klen > 1152:       # small plane, long reduction: ask the library whether K splits pay
        # reductions of 36-72 tap steps are split only when no host time is at stake: while a hipGraph is being captured
        short = klen <= 2304
        if not short or torch.cuda.is_current_stream_capturing():
            p.splitk = -1 if short else 0
            nsplit = _hip.load().ff_conv2d_splitk_hint(C.byref(p))
            p.splitk = 0
            if nsplit > 1:
                ws = torch.empty(nsplit * b * ho * wo * cout, dtype=torch.float32, device=x0.device)
                p.splitk_ws, p.splitk = ws.data_ptr(), nsplit
    nparts = _hip.load().ff_conv2d_stats_parts(C.byref(p)) if want_stats and CONV_STATS and not p.splitk else 0
    if nparts:
        parts = torch.empty(b * nparts * cout * 4, dtype=torch.float32, device=x0.device)
        p.stats_part = parts.data_ptr()
        _timed_call("conv", "ff_conv2d_fwd", C.byref(p), _stream(), note=(2.0 * b * ho * wo * cout * kh * kw * cin, w_fmt))
        stats = _zero_stats(b, cout, x0.device)
        _hip.call("ff_norm_stats_finish", _p(parts), b, nparts, cout, _p(stats), _stream())
        return out, stats
    _timed_call("conv", "ff_conv2d_fwd", C.byref(p), _stream(), note=(2.0 * b * ho * wo * cout * kh * kw * cin, w_fmt))
    if want_stats:
        return out, norm_stats(out, per_sample=True)
    if y_split is True:
        out = SplitT(out)
    if y2_split:
        return out, SplitT(out2)
    return out


class TiledPyramid:
    """The 4-level correlation pyramid in the tiled HBM layout of csrc/corr_layout.h.

    levels[l]: (B*Q, plane_l) tensor, float32 or float16; a plane is a grid of 128-byte tiles (8 x 4 floats or
    8 x 8 halfs) over the level's h_l x w_l values.  `rowmajor(l)` converts a level back to (B*Q, h_l, w_l) fp32
    (tests / debugging only)."""

    def __init__(self, levels: List[Tensor], h0: int, w0: int, half: bool):
        self.levels, self.h0, self.w0, self.half = levels, h0, w0, half

    @staticmethod
    def plane_elems(h0: int, w0: int, level: int, half: bool) -> int:
        n = _hip.load().ff_corr_plane_elems(h0, w0, level, int(half))
        if n <= 0:
            raise _hip.FocusFlowHipError(f"ff_corr_plane_elems({h0}, {w0}, {level}) failed")
        return n

    @classmethod
    def empty(cls, planes: int, h0: int, w0: int, half: bool, device, zero: bool = False) -> "TiledPyramid":
        # ONE allocation for the four levels (each starts on a 256-byte boundary): the lookup kernel addresses them
        # through a single buffer resource (csrc/corr_lookup_dma.hip), which needs them within 4 GB of each other
        dt = torch.float16 if half else torch.float32
        mk = torch.zeros if zero else torch.empty
        esz = 2 if half else 4
        n = [planes * cls.plane_elems(h0, w0, l, half) for l in range(4)]
        start, total = [], 0
        for l in range(4):
            start.append(total)
            total += (n[l] * esz + 255) // 256 * 256 // esz
        flat = mk((total,), dtype=dt, device=device)
        return cls([flat[start[l]:start[l] + n[l]].view(planes, n[l] // planes) for l in range(4)], h0, w0, half)

    def ptrs(self):
        return (C.c_void_p * 4)(*[lv.data_ptr() for lv in self.levels])

    def rowmajor(self, level: int) -> Tensor:
        n = self.levels[level].shape[0]
        out = torch.empty((n, self.h0 >> level, self.w0 >> level), dtype=torch.float32, device=self.levels[level].device)
        _hip.call("ff_corr_retile", _p(out), _p(self.levels[level]), n, self.h0, self.w0, level, int(self.half), 0, _stream())
        return out

    @classmethod
    def from_rowmajor(cls, levels: List[Tensor], half: bool = False) -> "TiledPyramid":
        """(B*Q, h_l, w_l) fp32 planes -> tiled storage (fp16: rounded to nearest even); pads are zero."""
        n, h0, w0 = levels[0].shape
        pyr = cls.empty(n, h0, w0, half, levels[0].device, zero=True)
        for l, lv in enumerate(levels):
            assert lv.shape == (n, h0 >> l, w0 >> l) and lv.is_contiguous() and lv.dtype == torch.float32
            _hip.call("ff_corr_retile", _p(lv), _p(pyr.levels[l]), n, h0, w0, l, int(half), 1, _stream())
        return pyr


def corr_build(fmap1: Tensor, fmap2: Tensor, half: bool = False) -> TiledPyramid:
    """CorrBlock.__init__ (corr.py:12-27, :52-60) in one launch: all-pairs volume / sqrt(C) on the f16 matrix pipe with
    fp16-split operands, the three 2x2 average-pooling levels from the accumulators, everything written once in the
    tiled layout (fp32, or fp16 storage = BASELINE configs[4]).  The exact-fp32 MFMA precision goes through the
    grouped-conv volume + pooling pass and is re-tiled."""
    b, h, w, c = fmap1.shape
    q = h * w
    assert fmap1.is_contiguous() and fmap2.is_contiguous() and fmap2.shape == fmap1.shape
    _require_gpu(fmap1)
    if w_format() == _hip.W_F32 or c != 256:       # (the one-term f16 mode builds the volume with the three-term kernel too)
        return TiledPyramid.from_rowmajor(corr_pyramid(corr_volume(fmap1, fmap2), h, w), half)
    esz = fmap1.element_size()
    if fmap2.data_ptr() == fmap1.data_ptr() + b * q * c * esz and fmap1.untyped_storage().data_ptr() == fmap2.untyped_storage().data_ptr():
        # both frames went through fnet as one batch: split them with one launch
        both = torch.empty((2 * b * q, c * 4), dtype=torch.uint8, device=fmap1.device)
        _hip.call("ff_pack_split_f16", _p(fmap1), _p(both), 2 * b * q, c, _stream())
        f1s, f2s = both[:b * q], both[b * q:]
    else:
        f1s, f2s = pack_split(fmap1.view(b * q, c)), pack_split(fmap2.view(b * q, c))
    pyr = TiledPyramid.empty(b * q, h, w, half, fmap1.device)
    _timed_call("corr_volume", "ff_corr_build", _p(f1s), _p(f2s), pyr.ptrs(), b, h, w, c, int(half), _stream())
    return pyr


def corr_lookup_tiled(pyr: TiledPyramid, coords: Tensor, want_taps: bool = False, out: Optional[Tensor] = None):
    """CorrBlock.__call__ (corr.py:29-50) on a TiledPyramid.  coords: (B, H, W, 2) [x, y].  Returns (B, H, W, 324)
    (+ int32 taps (B*H*W, 4, 2, 9)).  `out`: a (B,H,W,324) view of a wider buffer."""
    _require_gpu(coords)
    b, h, w, _ = coords.shape
    assert coords.is_contiguous() and pyr.levels[0].shape[0] == b * h * w
    if out is None:
        out = empty_nhwc(b, h, w, 324, coords)
    assert out.shape == (b, h, w, 324)
    taps = torch.empty((b * h * w, 4, 2, 9), dtype=torch.int32, device=coords.device) if want_taps else None
    _timed_call("lookup", "ff_corr_lookup_tiled_fwd", pyr.ptrs(), int(pyr.half), _p(coords), b * h * w, pyr.h0, pyr.w0,
                _p(out), _ld(out), _p(taps), _stream())
    return (out, taps) if want_taps else out


def corr_volume(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """corr.py:52-60: vol[b][i][j] = <f1[b,i,:], f2[b,j,:]> / sqrt(C) as a grouped 1x1 conv
    whose per-sample weights are fmap2.  Returns (B, Q, Q) planes [B*Q][H8][W8]."""
    b, h, w, c = fmap1.shape
    q = h * w
    assert fmap1.is_contiguous() and fmap2.is_contiguous() and fmap2.shape == fmap1.shape
    vol = torch.empty((b, q, q), dtype=torch.float32, device=fmap1.device)
    p = FFConvParams()
    p.x[0], p.x_ld[0], p.x_c[0], p.x_gstride[0] = fmap1.data_ptr(), c, c, q * c
    p.groups, p.B, p.H, p.W = b, 1, h, w
    p.w_format = w_format()
    if p.w_format == 0:
        p.w, p.w_gstride = fmap2.data_ptr(), q * c
    else:  # fmap2 plays the weights: split it once per pair
        f2s = pack_split(fmap2.view(b * q, c))
        p.w, p.w_gstride = f2s.data_ptr(), q * f2s.shape[1] // 4
    p.out_scale = 1.0 / math.sqrt(c)
    p.y, p.y_ld, p.y_gstride = vol.data_ptr(), q, q * q
    p.Ho, p.Wo, p.Cout = h, w, q
    p.KH = p.KW = p.stride = 1
    _require_gpu(fmap1)
    _timed_call("corr_volume", "ff_conv2d_fwd", C.byref(p), _stream())
    return vol


def corr_pyramid(vol: Tensor, h: int, w: int) -> List[Tensor]:
    b, q, _ = vol.shape
    n = b * q
    lv = [vol.view(n, h, w)]
    hh, ww = h, w
    for _ in range(3):
        hh, ww = hh // 2, ww // 2
        lv.append(torch.empty((n, hh, ww), dtype=torch.float32, device=vol.device))
    _hip.call("ff_corr_pyramid", _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), n, h, w, _stream())
    return lv


def corr_lookup(levels: List[Tensor], coords: Tensor, radius: int = 4, want_taps: bool = False,
                out: Optional[Tensor] = None):
    """coords: (B, H, W, 2) [x, y].  Returns (B, H, W, L*(2r+1)^2) (+ int32 taps).  `out`: a (B,H,W,nk) view of a
    wider buffer (the caller owns the channels beyond nk)."""
    _require_gpu(coords)
    b, h, w, _ = coords.shape
    assert coords.is_contiguous()
    nl = len(levels)
    nk = nl * (2 * radius + 1) ** 2
    if out is None:
        out = empty_nhwc(b, h, w, nk, coords)
    assert out.shape == (b, h, w, nk)
    taps = torch.empty((b * h * w, nl, 2, 2 * radius + 1), dtype=torch.int32, device=coords.device) if want_taps else None
    arr = (C.c_void_p * 4)(*[lv.data_ptr() for lv in levels] + [0] * (4 - nl))
    h0, w0 = levels[0].shape[-2:]
    _timed_call("lookup", "ff_corr_lookup_fwd", arr, nl, radius, _p(coords), b * h * w, h0, w0, _p(out), _ld(out), _p(taps), _stream())
    return (out, taps) if want_taps else out


# ----------------------------------------------------------------------------
# Zeroed fp64 statistics buffers come out of one arena per forward pass (one fill instead of one per norm layer).
# The arena is only handed out between begin_forward() and the next begin_forward(); every slice is used once.
_stats_arena = None
_stats_used = 0


def begin_forward(device):
    """Called by the model at the top of a forward pass: a fresh zeroed arena for the norm statistics."""
    global _stats_arena, _stats_used, _range_word
    _stats_arena = torch.zeros(1 << 18, dtype=torch.float64, device=device)     # 2 MB: ~60 norm layers x B x C x 2
    _stats_used = 0
    _range_word = torch.zeros(1, dtype=torch.int32, device=device) if CHECK_RANGE else None


# The split conv formats read an activation x as f16(4 x) + residual: |x| must stay below 16376 (csrc/ff_common.h), and a
# value beyond it becomes inf without any other sign than a NaN flow.  Weights are checked when they are loaded
# (model.load_state_dict); activations depend on the input, so their check is a debug mode: FF_CHECK_RANGE=1 (or
# ops.CHECK_RANGE = True) measures max|x| of every tensor a forward convolution reads (one extra read-only pass each, the
# kernel the backward uses for its gradient scales) and the model raises at the end of the forward pass that overflowed.
CHECK_RANGE = os.environ.get("FF_CHECK_RANGE", "0") == "1"
# InstanceNorm statistics from the producing convolution's epilogue (FFConvParams.stats_part) instead of a pass over its output
CONV_STATS = True
# ... and in recorded (training) passes: off - at the 46 x 62 planes of the training crop most tiles are ragged (the slow
# per-pixel branch of the epilogue) and the statistics pass it replaces is short: 70.3 vs 69.6 ms per step (A/B, round 3)
CONV_STATS_TRAIN = False
X_LIMIT = 16376.0
_range_word = None


def _range_probe(x: Tensor):
    b, h, w, c = x.shape
    if act_bwd_is_alias(x, ACT_NONE, 1.0, c):      # measures only, no copy
        _hip.call("ff_act_bwd", _p(x), _ld(x), None, 0, _p(x), c, b * h * w, c, c, ACT_NONE, 1.0, _p(_range_word), _stream())
    else:                                            # a channel slice / padded tensor: through a dense copy
        act_bwd(x, None, ACT_NONE, 1.0, c, want_amax=True, amax=_range_word)


def check_range(what: str = "forward pass"):
    """Debug mode (CHECK_RANGE): raise if an input of a split-format convolution of this pass left fp16's range."""
    if _range_word is None:
        return
    m = float(_range_word.view(torch.float32).item())       # the word holds the bits of max|x| (non-negative floats order as ints)
    if not (m < X_LIMIT):
        raise _hip.FocusFlowHipError(
            f"{what}: a convolution input reached |x| = {m:.6g}; the fp16-split conv formats need |x| < {X_LIMIT:g} "
            "(ff_common.h).  Run this checkpoint / input with FF_CONV_PRECISION=fp32.")


# ---- the always-on guard ---------------------------------------------------------------------------------------------------
# Two tensors per forward decide whether a checkpoint / input stays inside the split formats' range: the context encoder's
# output (no norm behind its last convolution: |x| ~ 700 on the synthetic test weights; slot 0) and the feature maps the
# correlation volume is built from (slot 1).  Their max|x| is measured in every forward (two read-only passes,
# ff_range_probe: ~15 us).  What happens with the two numbers depends on what is known about the model (`owner` = the RAFT
# module, which keeps the running maximum of what its forwards have shown):
#   * CAREFUL forwards - the first one after the weights changed, and every one while the running maximum is within a
#     factor 4 of the limit - read the words synchronously behind the encoders (one host sync, ~1 ms of lost run-ahead) and
#     act BEFORE the values meet a split convolution: a context output beyond the limit switches the module to its exact-fp32
#     route for the convolutions that read it (SepConvGRU.prepare: `_exact_ctx`, sticky, logged) - the flow of THAT forward is
#     right; feature maps beyond the limit have no local repair (the correlation values they produce overflow the next
#     layer as well) and raise.
#   * all other forwards copy the words to pinned host memory without a synchronisation and look at them when the NEXT
#     forward starts (or on demand: FF_RAFT_FUSION.check_range(), one host sync): a value that jumped past the limit from
#     below a quarter of it in one step raises FocusFlowHipError then - the flow of that pass was not valid - and arms the
#     exact route for what follows.
# The state lives in the per-thread policy object (two models driven from two threads do not see each other's words); hipGraph
# captures take a caller-provided static word (graph.GraphedForward) and are checked after each replay.
# Not covered: the encoders' INTERMEDIATE tensors and training tensors (debug mode FF_CHECK_RANGE=1).  FF_RANGE_GUARD=0: off.
RANGE_GUARD = os.environ.get("FF_RANGE_GUARD", "1") != "0"
GUARD_MARGIN = 4.0


def _guard_state():
    st = policy.__dict__.get("_guard")
    if st is None:
        st = policy.__dict__["_guard"] = {"words": None, "pending": [], "capture_words": None}
    return st


def guard_begin(device):
    st = _guard_state()
    st["words"] = None
    if torch.cuda.is_current_stream_capturing():      # (no event queries, host copies or fresh words inside a hipGraph capture)
        if RANGE_GUARD and st["capture_words"] is not None:
            st["words"] = st["capture_words"]
            st["words"].zero_()                         # (a captured fill: every replay starts from zero)
        return
    guard_check()
    if RANGE_GUARD:
        st["words"] = torch.zeros(2, dtype=torch.int32, device=device)


def guard_probe(x: Tensor, slot: int):
    """max|x| of an NHWC tensor into slot 0 (context encoder output) / 1 (feature maps) of the running forward's guard words."""
    w = _guard_state()["words"]
    if w is None:
        return
    b, h, ww, c = x.shape
    _hip.call("ff_range_probe", _p(x), _ld(x), b * h * ww, c, _p(w[slot:slot + 1]), _stream())


def guard_careful(owner) -> bool:
    """Should this forward read its guard words synchronously (see above)?"""
    st = _guard_state()
    if st["words"] is None or torch.cuda.is_current_stream_capturing():
        return False
    return (not getattr(owner, "_guard_hist", False)) or getattr(owner, "_guard_level", 0.0) * GUARD_MARGIN >= X_LIMIT


def guard_read_now():
    """(max|context output|, max|feature maps|) of the running forward so far: one host synchronisation."""
    w = _guard_state()["words"]
    a, b = w.view(torch.float32).tolist()
    return a, b


def guard_note(owner, m_ctx: float, m_fmap: float):
    """Fold one forward's maxima into the owner's running level."""
    m = max(m_ctx, m_fmap) if m_ctx == m_ctx and m_fmap == m_fmap else float("inf")
    owner._guard_level = max(getattr(owner, "_guard_level", 0.0), m)
    owner._guard_hist = True


def guard_end(what: str, owner=None, handled: bool = False):
    """End of a forward: queue its words for the asynchronous look (handled: a careful forward has read them already)."""
    st = _guard_state()
    w, st["words"] = st["words"], None
    if w is None or handled or torch.cuda.is_current_stream_capturing():
        return
    guard_queue(w, what, owner)


def guard_queue(words: Tensor, what: str, owner=None):
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    host.copy_(words, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _guard_state()["pending"].append((host, ev, what, owner))


def guard_check(sync: bool = False):
    """Look at the guard words of finished forwards (all of them if sync, else those whose copy has arrived)."""
    pending = _guard_state()["pending"]
    while pending:
        host, ev, what, owner = pending[0]
        if not ev.query():
            if not sync:
                return
            ev.synchronize()
        pending.pop(0)
        m_ctx, m_fmap = host.view(torch.float32).tolist()
        if owner is not None:
            guard_note(owner, m_ctx, m_fmap)
        ctx_bad = not (m_ctx < X_LIMIT) and not (getattr(owner, "_exact_ctx", False) and m_ctx < float("inf"))
        if ctx_bad or not (m_fmap < X_LIMIT):
            pending.clear()
            if owner is not None and ctx_bad and m_ctx < float("inf"):
                owner._exact_ctx = True            # the next forward of this model is right
            raise _hip.FocusFlowHipError(
                f"{what}: an encoder output reached |x| = {max(m_ctx, m_fmap):.6g}; the fp16-split conv formats need |x| < {X_LIMIT:g} "
                "(csrc/ff_common.h) - the flow of that pass is not valid.  "
                + ("The context features' convolutions run on the exact-fp32 route from now on." if ctx_bad and m_fmap < X_LIMIT and m_ctx < float("inf")
                   else "Run this checkpoint / input with FF_CONV_PRECISION=fp32."))


def _zero_stats(s, c, device):
    global _stats_used
    n = s * c * 2
    if _stats_arena is None or _stats_arena.device != device or _stats_used + n > _stats_arena.numel():
        return torch.zeros((s, c, 2), dtype=torch.float64, device=device)
    out = _stats_arena[_stats_used:_stats_used + n].view(s, c, 2)
    _stats_used += n
    return out


BATCH_STATS_PER_IMAGE = os.environ.get("FF_BATCH_STATS_PER_IMAGE", "1") != "0"      # A/B switch


def norm_stats(x: Tensor, per_sample: bool) -> Tensor:
    b, h, w, c = x.shape
    if not per_sample and b > 1 and BATCH_STATS_PER_IMAGE:
        # batch statistics: every block of the pass ends in one fp64 atomic pair per channel, and with ONE row of sums for the whole
        # batch those chains are B times longer than an InstanceNorm's (8 x 184 x 248 x 64: 75 us, of which the bytes are worth 20 -
        # tools/norm_lab.py).  Per-image rows + a sum over them costs one small launch more and keeps the chains short.
        part = _zero_stats(b, c, x.device)
        _hip.call("ff_norm_stats", _p(x), _ld(x), b, h * w, c, 1, _p(part), _stream())
        return part.sum(0, keepdim=True)
    stats = _zero_stats(b if per_sample else 1, c, x.device)
    _hip.call("ff_norm_stats", _p(x), _ld(x), b, h * w, c, int(per_sample), _p(stats), _stream())
    return stats


def norm_coeffs(stats: Tensor, count: int, eps: float = 1e-5, gamma=None, beta=None):
    """The (scale, shift) tables norm_apply would use, fp32 [S][C] each, for conv2d(in_scale=..., in_shift=...)."""
    s_, c, _ = stats.shape
    out = torch.empty((2, s_, c), dtype=torch.float32, device=stats.device)
    _hip.call("ff_norm_coeffs", _p(stats), s_, c, count, eps, _p(gamma), _p(beta), _p(out[0]), _p(out[1]), _stream())
    return out[0], out[1]


def norm_apply(x: Tensor, stats: Tensor, per_sample: bool, eps: float = 1e-5, gamma=None, beta=None,
               act: int = ACT_NONE, res: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    b, h, w, c = x.shape
    if out is None:
        out = empty_nhwc(b, h, w, c, x)
    _hip.call("ff_norm_apply", _p(x), _ld(x), _p(out), _ld(out), b, h * w, c, _p(stats), int(per_sample), eps,
              _p(gamma), _p(beta), act, _p(res), _ld(res) if res is not None else 0, _stream())
    return out


def bn_fold(bn: torch.nn.BatchNorm2d):
    """Eval-mode BatchNorm as per-channel (scale, shift) for the conv epilogue; cached until a parameter or
    running statistic changes (version counters / storage)."""
    key = tuple((t._version, t.data_ptr()) for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias))
    cached = getattr(bn, "_ff_fold", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    c = bn.num_features
    sc = torch.empty(c, dtype=torch.float32, device=bn.weight.device)
    sh = torch.empty_like(sc)
    _require_gpu(bn.weight)
    _hip.call("ff_bn_fold", _p(bn.running_mean), _p(bn.running_var), _p(bn.weight), _p(bn.bias), bn.eps, _p(sc),
              _p(sh), c, _stream())
    bn._ff_fold = (key, sc, sh)
    return sc, sh


def bn_update_running(bn: torch.nn.BatchNorm2d, stats: Tensor, count: int):
    _hip.call("ff_bn_update_running", _p(stats), count, bn.momentum, _p(bn.running_mean), _p(bn.running_var),
              bn.num_features, _stream())
    for t in (bn.running_mean, bn.running_var):     # written through raw pointers: tell PyTorch (bn_fold's cache key)
        torch.autograd.graph.increment_version(t)


# ----------------------------------------------------------------------------
def prep_input(src_nchw: Optional[Tensor], b, h, w, like: Tensor, fill: float = 0.0, out: Optional[Tensor] = None) -> Tensor:
    dst = empty_nhwc(b, h, w, 4, like) if out is None else out
    assert dst.shape == (b, h, w, 4) and dst.is_contiguous()
    if src_nchw is not None:
        _require_gpu(src_nchw)
        src_nchw = src_nchw.contiguous()
        assert src_nchw.shape[0] == b and src_nchw.shape[2:] == (h, w)
    _hip.call("ff_prep_input", _p(src_nchw), src_nchw.shape[1] if src_nchw is not None else 0, fill, _p(dst), b, h, w,
              _stream())
    return dst


def cat_batch(a: Tensor, b: Tensor) -> Tensor:
    """torch.cat([a, b], 0) - without the copy when b lies right behind a in one buffer (the model prepares both frames of a
    pair into the two halves of one tensor, so that the feature encoder sees them as ONE batch of 2B, raft.py:187-189)."""
    if (a.is_contiguous() and b.is_contiguous() and a.shape[1:] == b.shape[1:] and a.dtype == b.dtype
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.storage_offset() + a.numel()):
        return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], 0)


def mask_prepare(mode: int, mask_nchw: Tensor, image: Tensor, table: Tensor, raw: bool = False, image_nhwc4: bool = False) -> Tensor:
    """init_mask modes neighborG(0)/neighborE(1)/context(2) -> NHWC4; scaled to [-1,1] (FF-RAFT) or raw [0,255] (FF-PWC);
    the context image is NCHW (B,3,H,W) or, with image_nhwc4, an NHWC4 tensor."""
    _require_gpu(mask_nchw)
    b, _, h, w = mask_nchw.shape
    dst = empty_nhwc(b, h, w, 4, mask_nchw)
    tmp = torch.empty((b, h, w), dtype=torch.float32, device=mask_nchw.device)
    gmax = torch.empty(1, dtype=torch.int32, device=mask_nchw.device)
    _hip.call("ff_mask_prepare", mode | (4 if raw else 0) | (8 if image_nhwc4 else 0), _p(mask_nchw.contiguous()), _p(image.contiguous()),
              _p(table), table.shape[0], _p(tmp), _p(gmax), _p(dst), b, h, w, _stream())
    return dst


# ----------------------------------------------------------------------------
# SA / CA fusion units (parallel_fusion.py:14-73)
# ----------------------------------------------------------------------------
SPATIAL_SLABS = 64   # FF_SPATIAL_SLABS


def chan_stats(x: Tensor):
    """(B,H,W,C) -> ((B,H,W,4) = [mean_c, max_c, 0, 0], argmax (B,H,W) int32)."""
    _require_gpu(x)
    b, h, w, c = x.shape
    st = empty_nhwc(b, h, w, 4, x)
    am = torch.empty((b, h, w), dtype=torch.int32, device=x.device)
    _hip.call("ff_chan_stats_fwd", _p(x), _ld(x), c, b * h * w, _p(st), 4, _p(am), _stream())
    return st, am


def chan_stats_bwd(g: Tensor, argmax: Tensor, c: int) -> Tensor:
    b, h, w, _ = g.shape
    gx = empty_nhwc(b, h, w, c, g)
    _hip.call("ff_chan_stats_bwd", _p(g), _ld(g), _p(argmax), c, b * h * w, _p(gx), _ld(gx), _stream())
    return gx


def _spatial_scratch(b, c, like):
    return torch.empty(SPATIAL_SLABS * b * c * 3, dtype=torch.float32, device=like.device)


def spatial_stats(x: Tensor):
    """(B,H,W,C) -> ((2B,1,1,C): rows [0,B) = mean over pixels, rows [B,2B) = max; argmax (B,C) int32)."""
    _require_gpu(x)
    b, h, w, c = x.shape
    out = torch.empty((2 * b, 1, 1, c), dtype=torch.float32, device=x.device)
    am = torch.empty((b, c), dtype=torch.int32, device=x.device)
    _hip.call("ff_spatial_stats_fwd", _p(x), _ld(x), c, b, h * w, _p(out), _p(out[b:]), _p(am), _p(_spatial_scratch(b, c, x)),
              _stream())
    return out, am


def spatial_stats_bwd(g: Tensor, argmax: Tensor, h: int, w: int) -> Tensor:
    b2, _, _, c = g.shape
    b = b2 // 2
    g = g.contiguous()
    gx = empty_nhwc(b, h, w, c, g)
    _hip.call("ff_spatial_stats_bwd", _p(g), _p(g[b:]), _p(argmax), c, b, h * w, _p(gx), _ld(gx), _stream())
    return gx


def scale_add(v: Tensor, s: Tensor, q: Optional[Tensor], mode: int) -> Tensor:
    """out = s * v + q.  mode 0: s (B,H,W,1) per pixel; mode 1: s (2B,1,1,C), scale = s[:B] + s[B:]."""
    _require_gpu(v)
    b, h, w, c = v.shape
    out = empty_nhwc(b, h, w, c, v)
    if mode == 0:
        assert s.shape[:3] == (b, h, w)
        _hip.call("ff_scale_add_fwd", _p(v), _ld(v), _p(s), _ld(s), None, _p(q), _ld(q) if q is not None else 0, _p(out),
                  _ld(out), c, b, h * w, 0, _stream())
    else:
        assert s.shape == (2 * b, 1, 1, c) and s.is_contiguous()
        _hip.call("ff_scale_add_fwd", _p(v), _ld(v), _p(s), 0, _p(s[b:]), _p(q), _ld(q) if q is not None else 0, _p(out),
                  _ld(out), c, b, h * w, 1, _stream())
    return out


def scale_add_bwd(gout: Tensor, v: Tensor, s: Tensor, mode: int):
    """-> (gv, gs) with gs shaped like s."""
    b, h, w, c = v.shape
    gv = empty_nhwc(b, h, w, c, v)
    if mode == 0:
        gs = torch.empty((b, h, w, 1), dtype=torch.float32, device=v.device)
        _hip.call("ff_scale_add_bwd", _p(gout), _ld(gout), _p(v), _ld(v), _p(s), _ld(s), None, _p(gv), _ld(gv), _p(gs), None,
                  c, b, h * w, 0, _stream())
    else:
        gs = torch.empty((2 * b, 1, 1, c), dtype=torch.float32, device=v.device)
        _hip.call("ff_scale_add_bwd", _p(gout), _ld(gout), _p(v), _ld(v), _p(s), 0, _p(s[b:]), _p(gv), _ld(gv), _p(gs),
                  _p(_spatial_scratch(b, c, v)), c, b, h * w, 1, _stream())
    return gv, gs


def act_copy(src: Tensor, dst: Tensor, act: int):
    b, h, w, c = src.shape
    assert dst.shape == src.shape
    _hip.call("ff_act_copy", _p(src), _ld(src), _p(dst), _ld(dst), b * h * w, c, act, _stream())


def coords_init(b, h, w, like: Tensor, flow_init: Optional[Tensor] = None) -> Tensor:
    coords = empty_nhwc(b, h, w, 2, like)
    if flow_init is not None:
        _require_gpu(flow_init)
        flow_init = flow_init.contiguous()
        assert flow_init.shape == (b, 2, h, w)
    _hip.call("ff_coords_init", _p(coords), _p(flow_init), b, h, w, _stream())
    return coords


def coords_step(coords1: Tensor, delta: Optional[Tensor], flow4: Optional[Tensor], slot: Optional[Tensor]):
    b, h, w, _ = coords1.shape
    _hip.call("ff_coords_step", _p(coords1), _p(delta), _ld(delta) if delta is not None else 0, _p(flow4), _p(slot),
              _ld(slot) if slot is not None else 0, b, h, w, _stream())


def gru_rh(r: Tensor, h: Tensor, out: Optional[Tensor] = None) -> Tensor:
    b, hh, ww, c = h.shape
    if out is None:
        out = empty_nhwc(b, hh, ww, c, h)
    _hip.call("ff_gru_rh", _p(r), _ld(r), _p(h), _ld(h), _p(out), _ld(out), b * hh * ww, c, _stream())
    return out


def gru_blend(z: Tensor, q: Tensor, h: Tensor, out: Optional[Tensor] = None) -> Tensor:
    b, hh, ww, c = h.shape
    if out is None:
        out = empty_nhwc(b, hh, ww, c, h)
    _hip.call("ff_gru_blend", _p(z), _ld(z), _p(q), _ld(q), _p(h), _ld(h), _p(out), _ld(out), b * hh * ww, c, _stream())
    return out


def gru_pass(direction: int, hs, motion, h: Tensor, zr_pre: Tensor, q_pre: Tensor, wzr_frag: Tensor, wq_frag: Tensor, bzr: Tensor, bq: Tensor,
             w_fmt: int, y: Optional[Tensor] = None, y2: Optional[Tensor] = None, gates=None):
    """One SepConvGRU pass as one launch (ff_gru_pass): hs / motion SplitT, h fp32 -> (h' fp32, h' SplitT).  y / y2: existing
    outputs; gates = (z, r, q) fp32 (B,H,W,128) tensors of one leading dimension: the recorded form (ff_gru_pass_rec) also
    stores what the backward differentiates through."""
    b, hh, ww, c = h.shape
    assert c == 128 and hs.shape == h.shape and motion.shape == h.shape and zr_pre.shape == (b, hh, ww, 256) and q_pre.shape == h.shape
    y = empty_nhwc(b, hh, ww, c, h) if y is None else y
    y2 = empty_nhwc(b, hh, ww, c, h) if y2 is None else y2
    note = (2.0 * b * hh * ww * 384 * 384 * 5, w_fmt)
    # (timed with the convolutions in bench.py's roofline_conv: it IS two of them - 384 -> 256 and 384 -> 128, five taps)
    if gates is None:
        _timed_call("conv", "ff_gru_pass", direction, _p(hs.t), _ld(hs.t), _p(motion.t), _ld(motion.t), _p(h), _ld(h), _p(zr_pre), _ld(zr_pre), _p(q_pre), _ld(q_pre),
                    _p(wzr_frag), _p(wq_frag), _p(bzr), _p(bq), w_fmt, _p(y), _ld(y), _p(y2), _ld(y2), b, hh, ww, _stream(), note=note)
    else:
        z, r, q = gates
        assert z.shape == r.shape == q.shape == h.shape and _ld(z) == _ld(r) == _ld(q)
        _timed_call("conv", "ff_gru_pass_rec", direction, _p(hs.t), _ld(hs.t), _p(motion.t), _ld(motion.t), _p(h), _ld(h), _p(zr_pre), _ld(zr_pre), _p(q_pre), _ld(q_pre),
                    _p(wzr_frag), _p(wq_frag), _p(bzr), _p(bq), w_fmt, _p(y), _ld(y), _p(y2), _ld(y2), _p(z), _p(r), _p(q), _ld(z), b, hh, ww, _stream(), note=note)
    return y, SplitT(y2)


class LazyAct:
    """An activation that exists only as its ingredients: relu(InstanceNorm(t)) after a stem, relu(x + relu(InstanceNorm(t)))
    at the end of a residual stage - `t` raw convolution output, `scale` / `shift` the [B][C] tables of norm_coeffs, `res` the
    block's input or None.  The one consumer that can read it like this is fusion_pair (csrc/fusion_pair.hip: the
    normalisation pass, its write and its re-read disappear); anything else calls materialise() = ff_norm_apply's result,
    bit for bit."""
    __slots__ = ("t", "stats", "count", "eps", "act", "res", "_coef")

    def __init__(self, t: Tensor, stats: Tensor, count: int, eps: float, act: int, res: Optional[Tensor]):
        self.t, self.stats, self.count, self.eps, self.act, self.res, self._coef = t, stats, count, eps, act, res, None

    @property
    def shape(self):
        return self.t.shape

    def coeffs(self):
        if self._coef is None:
            self._coef = norm_coeffs(self.stats, self.count, self.eps)
        return self._coef

    def materialise(self) -> Tensor:
        return norm_apply(self.t, self.stats, True, self.eps, act=self.act, res=self.res, out=self.t)

    def record_stream(self, stream):
        for x in (self.t, self.stats, self.res) + (tuple(self._coef) if self._coef is not None else ()):
            if x is not None:
                x.record_stream(stream)


def fusion_pair_tile(c: int) -> int:
    """Pixels per tile of ff_fusion_pair_fwd for c channels per branch (0: no instance)."""
    return _hip.load().ff_fusion_pair_tile(c)


def fusion_pair_inputs_ok(*ins) -> bool:
    """ff_fusion_pair_fwd reads contiguous NHWC tensors (leading dimension == C) of less than 2 GiB."""
    for v in ins:
        for t in ((v.t, v.res) if isinstance(v, LazyAct) else (v,)):
            if t is not None and (not t.is_cuda or _ld(t) != t.shape[3] or t.numel() * 4 >= (1 << 31)):
                return False
    return True


def fusion_pair(img, mask, w_frag, bias, w_fmt: int):
    """FusionUnit '1x1conv', both directions, one launch (ff_fusion_pair_fwd): img / mask fp32 NHWC tensors or LazyAct;
    w_frag = (mask2img, img2mask) weights in fragment order, bias likewise -> (img', mask')."""
    ins = (img, mask)
    b, h, w, c = ins[0].shape
    assert ins[1].shape == ins[0].shape
    p = _hip.FFFusionPair()
    outs = []
    for i, v in enumerate(ins):
        t = v.t if isinstance(v, LazyAct) else v
        _require_gpu(t)
        p.x[i], p.x_ld[i] = t.data_ptr(), _ld(t)
        if isinstance(v, LazyAct):
            sc, sh = v.coeffs()
            p.scale[i], p.shift[i] = sc.data_ptr(), sh.data_ptr()
            p.in_act = v.act
            if v.res is not None:
                p.xres[i], p.xres_ld[i] = v.res.data_ptr(), _ld(v.res)
        y = t if isinstance(v, LazyAct) else empty_nhwc(b, h, w, c, t)       # a lazy input's raw tensor has no other reader: write over it
        outs.append(y)
        p.y[i], p.y_ld[i] = y.data_ptr(), _ld(y)
        p.w_frag[i] = w_frag[i].data_ptr()
        p.bias[i] = bias[i].data_ptr() if bias[i] is not None else None
    if _range_word is not None:      # CHECK_RANGE: what the unit reads goes through the split format like any convolution input
        for v in ins:
            # (a lazy value is relu(res + relu(InstanceNorm(t))): bounded by the residual, which is probed; t itself never meets the format)
            probe = v.res if isinstance(v, LazyAct) else v
            if probe is not None:
                _range_probe(probe)
    acts = {v.act for v in ins if isinstance(v, LazyAct)}
    assert len(acts) <= 1, "lazy inputs of one fusion unit share the activation"
    p.w_format, p.B, p.HW, p.C = w_fmt, b, h * w, c
    # (timed with the convolutions in bench.py's roofline_conv: two C x C 1x1 convolutions)
    _timed_call("conv", "ff_fusion_pair_fwd", C.byref(p), _stream(), note=(2.0 * b * h * w * 2 * c * c, w_fmt))
    return outs[0], outs[1]


def mask_upsample_pack(w_split: Tensor) -> Tensor:
    """Split rows of the mask head's second convolution (576 x 256) -> ff_mask_upsample_fwd's stage-major weight image."""
    assert w_split.dtype == torch.uint8 and w_split.numel() == 576 * 1024 and w_split.is_contiguous()
    out = torch.empty_like(w_split)
    _hip.call("ff_mask_upsample_pack", _p(w_split), _p(out), _stream())
    return out


def mask_upsample(hid: Tensor, w_stage: Tensor, w_fmt: int, bias: Optional[Tensor], flow: Tensor, out_scale: float = 0.25) -> Tensor:
    """mask[2] (1x1, 256 -> 576) * out_scale + soft-max + convex up-sampling in one launch (ff_mask_upsample_fwd).
    hid: (B,H,W,256) view of the mask head's hidden tensor; w_stage: mask_upsample_pack(split rows); flow: (B,H,W,>=2).
    -> (B,2,8H,8W)."""
    b, h, w, c = hid.shape
    assert c == 256 and w_fmt in (_hip.W_F16X3, _hip.W_F16) and flow.shape[:3] == (b, h, w)
    out = torch.empty((b, 2, 8 * h, 8 * w), dtype=torch.float32, device=hid.device)
    _hip.call("ff_mask_upsample_fwd", _p(hid), _ld(hid), _p(w_stage), w_fmt, _p(bias), out_scale, _p(flow), _ld(flow), _p(out), b, h, w, _stream())
    return out


def upsample_flow(flow: Tensor, mask: Tensor) -> Tensor:
    """flow (B,H,W,>=2) NHWC, mask (B,H,W,576) NHWC -> (B,2,8H,8W) NCHW."""
    b, h, w, _ = flow.shape
    out = torch.empty((b, 2, 8 * h, 8 * w), dtype=torch.float32, device=flow.device)
    _hip.call("ff_upsample_flow", _p(flow), _ld(flow), _p(mask), _ld(mask), _p(out), b, h, w, _stream())
    return out


def nhwc_to_nchw(x: Tensor) -> Tensor:
    b, h, w, c = x.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    _hip.call("ff_nhwc_to_nchw", _p(x), _ld(x), _p(out), b, h, w, c, _stream())
    return out


# ----------------------------------------------------------------------------
# backward wrappers
# ----------------------------------------------------------------------------
def _conv_params(xs, b, h, w, cout, kh, kw, stride, pad, ho, wo, dilation=1):
    p = FFConvParams()
    p.dil_h = p.dil_w = dilation
    for i, x in enumerate(xs):
        p.x[i], p.x_ld[i], p.x_c[i], p.x_gstride[i] = x.data_ptr(), _ld(x), x.shape[3], 0
    p.groups, p.B, p.H, p.W = 1, b, h, w
    p.Ho, p.Wo, p.Cout = ho, wo, cout
    p.KH, p.KW, p.stride, p.pad_h, p.pad_w = kh, kw, stride, pad[0], pad[1]
    p.out_scale = 1.0
    return p


def conv2d_wgrad(xs: Sequence[Tensor], g: Tensor, cout: int, kh: int, kw: int, stride: int, pad,
                 g_amax: Optional[Tensor] = None, want_db: bool = False, dw: Optional[Tensor] = None,
                 db: Optional[Tensor] = None, dilation: int = 1):
    """packed dW [cout][kh*kw*cin] from inputs `xs` and output gradient g (B,Ho,Wo,>=cout, ld % 4 == 0).
    With g_amax (bits of max|g|, act_bwd) and a split conv format the kernel runs on the f16 matrix pipe and can
    also return the bias gradient: -> dW, or (dW, db) when want_db.  `dw` / `db`: existing buffers to ADD into
    (the kernel accumulates with atomics; fresh zeroed ones are allocated when omitted)."""
    b, h, w, _ = xs[0].shape
    _, ho, wo, _ = g.shape
    cin = sum(x.shape[3] for x in xs)
    if dw is None:
        dw = torch.zeros((cout, kh * kw * cin), dtype=torch.float32, device=g.device)
    assert dw.shape == (cout, kh * kw * cin) and dw.is_contiguous()
    p = _conv_params(xs, b, h, w, cout, kh, kw, stride, pad, ho, wo, dilation)
    p.y, p.y_ld = g.data_ptr(), _ld(g)
    fmt = w_format() if g_amax is not None else 0
    if fmt:
        p.w_format, p.x_amax = fmt, g_amax.data_ptr()
        if want_db and db is None:
            db = torch.zeros(cout, dtype=torch.float32, device=g.device)
    else:
        assert db is None, "accumulating bias-gradient buffers need a split conv format"
    _hip.call("ff_conv2d_wgrad", C.byref(p), _p(dw), 0, _p(db) if fmt else None, _stream())
    if want_db:
        return dw, (db if db is not None else channel_sum(g, cout))
    return dw


def unpack_conv_wgrad(packed: Tensor, cout, cin, kh, kw, cin_pad, cout_offset) -> Tensor:
    dw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=packed.device)
    _hip.call("ff_unpack_conv_wgrad", _p(packed), cout, cin, kh, kw, cin_pad, cout_offset, _p(dw), _stream())
    return dw


def pack_weights_table(table_dev: Tensor, njobs: int, total_blocks: int):
    """Run a device-resident FFPackJob table (cce.prepack builds it): every stale weight layout in one launch."""
    if not table_dev.is_cuda:
        raise _hip.FocusFlowHipError("ff_pack_weights_table: the job table must live on the HIP device")
    _hip.call("ff_pack_weights_table", _p(table_dev), njobs, total_blocks, _stream())


def unpack_wgrad_group(dw: Tensor, db: Optional[Tensor], couts, offs, has_bias, cin_src: int, slices, kh: int, kw: int,
                       cin_pad: int) -> Tensor:
    """Packed dW rows (+ db) of one PackedConv -> one flat tensor: per member its OIHW gradient (zero outside `slices`),
    then its bias gradient if has_bias[m]."""
    _require_gpu(dw)
    n = len(couts)
    total = sum(c * cin_src * kh * kw + (c if hb else 0) for c, hb in zip(couts, has_bias))
    out = torch.empty(total, dtype=torch.float32, device=dw.device)
    arr = C.c_int * n
    ns = len(slices) if slices else 0
    sarr = C.c_int * max(ns, 1)
    _hip.call("ff_unpack_wgrad_group", _p(dw), _p(db), n, arr(*couts), arr(*offs), arr(*[int(bool(h)) for h in has_bias]), cin_src, ns,
              sarr(*([lo for lo, _ in slices] if ns else [0])), sarr(*([hi for _, hi in slices] if ns else [0])), kh, kw, cin_pad,
              _p(out), _stream())
    return out


def pack_conv_weight_dgrad(w_oihw: Tensor, dst: Tensor, cout_pad: int, cout_offset: int):
    co, ci, kh, kw = w_oihw.shape
    # rows beyond Cin (channel padding of the forward input) stay zero: their input gradient is zero
    assert dst.is_contiguous() and dst.shape[0] >= ci and dst.shape[1] == kh * kw * cout_pad
    _hip.call("ff_pack_conv_weight_dgrad", _p(w_oihw.contiguous()), co, ci, kh, kw, _p(dst), cout_pad, cout_offset, _stream())


def act_bwd(dy: Tensor, y: Optional[Tensor], act: int, scale: float, c: int, want_amax: bool = False,
            amax: Optional[Tensor] = None):
    """g = dy*act'(y)*scale over the first c channels, zero-padded to a multiple of 4.
    want_amax: also return a device word with the bits of max|g| (-> conv2d(x_amax=...)); `amax`: a zeroed int32
    word to use for it instead of allocating one."""
    b, h, w, _ = dy.shape
    cpad = (c + 3) // 4 * 4
    # no activation, no scale, nothing to pad: g IS dy (ff_act_bwd with g == dy only measures max|dy|, no copy)
    alias = act == ACT_NONE and scale == 1.0 and dy.shape[3] == c == cpad and _ld(dy) == cpad and dy.data_ptr() % 16 == 0
    g = dy if alias else empty_nhwc(b, h, w, cpad, dy)
    if alias and not want_amax:
        return g
    if want_amax and amax is None:
        amax = torch.zeros(1, dtype=torch.int32, device=dy.device)
    _hip.call("ff_act_bwd", _p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(g), cpad, b * h * w, c, cpad,
              act, scale, _p(amax), _stream())
    return (g, amax) if want_amax else g


def act_bwd_into(dy: Tensor, y: Optional[Tensor], act: int, g: Tensor, amax: Tensor, scale: float = 1.0):
    """ff_act_bwd into an existing gradient tensor g (same channel count as dy, a multiple of 4), max|g| into the zeroed
    word `amax` (the recorded update loop: train_loop.py)."""
    b, h, w, c = dy.shape
    assert g.shape == dy.shape and c % 4 == 0
    _hip.call("ff_act_bwd", _p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(g), _ld(g), b * h * w, c, c, act, scale, _p(amax), _stream())
    return g


def deconv4x4s2_small(x: Tensor, w_rows: Tensor, bias: Optional[Tensor], cout: int, out: Tensor) -> Tensor:
    """ConvTranspose2d(Cin, cout <= 2, 4, 2, 1) of NHWC x (B,H,W,Cin) into `out` (B,2H,2W,>=cout) - ff_deconv4x4s2_small.
    w_rows: fp32 rows [cout][16*Cin] of the equivalent forward conv (transposed + flipped parameter, pack_conv_weight)."""
    b, h, w, cin = x.shape
    assert out.shape[:3] == (b, 2 * h, 2 * w) and w_rows.shape == (cout, 16 * cin) and w_rows.dtype == torch.float32
    _hip.call("ff_deconv4x4s2_small", _p(x), _ld(x), b, h, w, cin, _p(w_rows), _p(bias), cout, _p(out), _ld(out), _stream())
    return out


def dilate2(g: Tensor, hd: int, wd: int) -> Tensor:
    b, ho, wo, c = g.shape
    out = empty_nhwc(b, hd, wd, c, g)
    _hip.call("ff_dilate2", _p(g), _ld(g), _p(out), b, ho, wo, hd, wd, c, _stream())
    return out


def channel_sum(g: Tensor, c: int) -> Tensor:
    """sum over pixels of the first c channels (bias gradient); fp64 accumulation in ff_norm_stats."""
    b, h, w, cp = g.shape
    outs = []
    for lo in range(0, c, 256):          # ff_norm_stats handles <= 256 channels per call
        hi = min(cp, lo + 256)
        st = norm_stats(g[..., lo:hi], per_sample=False)
        outs.append(st[0, :, 0])
    return torch.cat(outs)[:c].float()


def act_bwd_is_alias(dy: Tensor, act: int, scale: float, c: int) -> bool:
    """act_bwd's no-copy case: no activation, no scale, nothing to pad - the conv's gradient IS dy."""
    return act == ACT_NONE and scale == 1.0 and dy.shape[3] == c == (c + 3) // 4 * 4 and _ld(dy) == c and dy.data_ptr() % 16 == 0


def norm_bwd(x, dy, y, fstats, per_sample, fixed_stats, eps, gamma, beta, relu, want_dres, amax: Optional[Tensor] = None,
             bstats: Optional[Tensor] = None):
    """amax: a zeroed int32 word that receives the bits of max|dx| (saves the consumer conv's measuring pass);
    bstats: a zeroed fp64 (S, C, 2) buffer for the two backward sums (allocated here when omitted)."""
    b, h, w, c = x.shape
    if bstats is None:
        bstats = torch.zeros((b if per_sample else 1, c, 2), dtype=torch.float64, device=x.device)
    assert bstats.shape == (b if per_sample else 1, c, 2) and bstats.dtype == torch.float64
    dx = empty_nhwc(b, h, w, c, x)
    dres = empty_nhwc(b, h, w, c, x) if want_dres else None
    _hip.call("ff_norm_bwd", _p(x), _ld(x), _p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(fstats),
              _p(bstats), int(per_sample), int(fixed_stats), eps, _p(gamma), _p(beta), int(relu), _p(dx), c,
              _p(dres), c if want_dres else 0, b, h * w, c, _p(amax), _stream())
    return dx, dres, bstats


def corr_lookup_tiled_bwd(dpyr: TiledPyramid, coords: Tensor, dout: Tensor):
    """Scatter d(lookup output) into the tiled fp32 gradient planes (accumulates)."""
    b, h, w, _ = coords.shape
    assert not dpyr.half
    _hip.call("ff_corr_lookup_tiled_bwd", dpyr.ptrs(), _p(coords), _p(dout), _ld(dout), b * h * w, dpyr.h0, dpyr.w0, _stream())


LOOKUP_BWD_ALL_MAX = 32      # csrc/corr_lookup_tiled.hip: LBA_MAXT


def lookup_bwd_all_fits(h0: int, w0: int) -> bool:
    """Whether ff_corr_lookup_tiled_bwd_all takes planes of this size (the four gradient planes of a query in 64 KB of LDS)."""
    return sum((h0 >> l) * (w0 >> l) for l in range(4)) * 4 <= 64 * 1024


def corr_lookup_tiled_bwd_all(coords_list, dout_list, h0: int, w0: int):
    """d(volume) [B*Q][plane_0] (tiled fp32) from every lookup of a pass at once (ff_corr_lookup_tiled_bwd_all), or None
    when the kernel declines (planes too large for LDS, more than 32 lookups): the caller then goes launch by launch."""
    n = len(coords_list)
    b, h, w, _ = coords_list[0].shape
    q = b * h * w
    ld = _ld(dout_list[0])
    assert all(c.is_contiguous() and c.shape == (b, h, w, 2) for c in coords_list) and all(_ld(d) == ld for d in dout_list)
    d0 = torch.empty((q, TiledPyramid.plane_elems(h0, w0, 0, False)), dtype=torch.float32, device=coords_list[0].device)
    cl = (C.c_void_p * n)(*[c.data_ptr() for c in coords_list])
    dl = (C.c_void_p * n)(*[d.data_ptr() for d in dout_list])
    lib = _hip.load()
    rc = lib.ff_corr_lookup_tiled_bwd_all(_p(d0), cl, dl, n, ld, q, h0, w0, _stream())
    if rc == 1:
        return None
    if rc != 0:
        raise _hip.FocusFlowHipError(f"ff_corr_lookup_tiled_bwd_all failed ({rc}): {lib.ff_last_error().decode()}")
    return d0


def corr_pyramid_tiled_bwd(dpyr: TiledPyramid):
    """avg_pool2d backward chain, in place: afterwards dpyr.levels[0] is d(volume) in tile order."""
    lv = dpyr.levels
    _hip.call("ff_corr_pyramid_tiled_bwd", _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), lv[0].shape[0], dpyr.h0, dpyr.w0, _stream())


def corr_tile_rows(x: Tensor, h: int, w: int, to_tiled: bool) -> Tensor:
    """(B, Q, C) feature rows <-> (B, P, C) rows in the tile order of a level-0 plane (pad rows zero)."""
    b, n, c = x.shape
    p = TiledPyramid.plane_elems(h, w, 0, False)
    assert x.is_contiguous() and n == (h * w if to_tiled else p)
    out = torch.empty((b, p if to_tiled else h * w, c), dtype=torch.float32, device=x.device)
    _hip.call("ff_corr_tile_rows", _p(x), _p(out), b, h, w, c, int(to_tiled), _stream())
    return out


def grouped_1x1(x: Tensor, wt: Tensor, out_scale: float) -> Tensor:
    """y[g] = x[g] (M x K) @ wt[g]^T (N x K)^T * out_scale, all contiguous (G, M, K), (G, N, K)."""
    g, m, k = x.shape
    n = wt.shape[1]
    y = torch.empty((g, m, n), dtype=torch.float32, device=x.device)
    p = FFConvParams()
    p.x[0], p.x_ld[0], p.x_c[0], p.x_gstride[0] = x.data_ptr(), k, k, m * k
    p.groups, p.B, p.H, p.W = g, 1, 1, m
    p.w, p.w_gstride = wt.data_ptr(), n * k
    p.out_scale = out_scale
    p.y, p.y_ld, p.y_gstride = y.data_ptr(), n, m * n
    p.Ho, p.Wo, p.Cout = 1, m, n
    p.KH = p.KW = p.stride = 1
    _require_gpu(x)
    _hip.call("ff_conv2d_fwd", C.byref(p), _stream())
    return y


def corr_volume_bwd(dvol: Tensor, f1: Tensor, f2: Tensor, tiled: bool = False):
    """BmmBackward of corr.py:58: df1 = dvol @ f2 / sqrt(C), df2 = dvol^T @ f1 / sqrt(C).
    tiled: dvol is (B, Q, P) with the fmap2 index j in the tile order of a level-0 plane (pad columns zero): the
    contraction over j runs over fmap2's rows gathered into the same order, and df2's rows are un-permuted at the end."""
    b, h, w, c = f1.shape
    q = h * w
    s = 1.0 / math.sqrt(c)
    if tiled:
        qn = dvol.shape[-1]
        dvol = dvol.contiguous().view(b, q, qn)
        f2r = corr_tile_rows(f2.contiguous().view(b, q, c), h, w, True)        # (B, P, C), pad rows zero
        qp = qn
    else:
        qn = q
        dvol = dvol.contiguous().view(b, q, q)
        qp = (q + 3) // 4 * 4
        if qp != q:      # odd plane sizes: the kernels read the contraction index in 16-byte groups -> zero-pad its rows
            dvol = torch.nn.functional.pad(dvol, (0, qp - q))
        f2r = f2.contiguous().view(b, q, c)
    if qp == qn:
        f2t = f2r.transpose(1, 2).contiguous()                                    # [c][j] = f2[j][c]: one launch for the batch
    else:
        f2t = torch.zeros((b, c, qp), dtype=torch.float32, device=f1.device)
        for i in range(b):
            _hip.call("ff_pack_conv_weight_dgrad", _p(f2r[i]), qn, c, 1, 1, _p(f2t[i]), qp, 0, _stream())
    df1 = grouped_1x1(dvol, f2t, s).view(b, h, w, c)
    df2 = torch.zeros((b, qn, c), dtype=torch.float32, device=f1.device)
    p = FFConvParams()
    p.x[0], p.x_ld[0], p.x_c[0], p.x_gstride[0] = f1.data_ptr(), c, c, q * c
    p.groups, p.B, p.H, p.W = b, 1, 1, q
    p.Ho, p.Wo, p.Cout = 1, q, qn
    p.KH = p.KW = p.stride = 1
    p.out_scale = s
    p.y, p.y_ld, p.y_gstride = dvol.data_ptr(), qp, q * qp
    _hip.call("ff_conv2d_wgrad", C.byref(p), _p(df2), qn * c, None, _stream())
    if tiled:
        df2 = corr_tile_rows(df2, h, w, False)
    return df1, df2.view(b, h, w, c)


def gru_rh_bwd(drh, r, h):
    b, hh, ww, c = h.shape
    dr, dh = empty_nhwc(b, hh, ww, c, h), empty_nhwc(b, hh, ww, c, h)
    _hip.call("ff_gru_rh_bwd", _p(drh), _ld(drh), _p(r), _ld(r), _p(h), _ld(h), _p(dr), c, _p(dh), c, b * hh * ww, c, _stream())
    return dr, dh


def gru_blend_bwd(dhn, z, q, h):
    b, hh, ww, c = h.shape
    dz, dq, dh = (empty_nhwc(b, hh, ww, c, h) for _ in range(3))
    _hip.call("ff_gru_blend_bwd", _p(dhn), _ld(dhn), _p(z), _ld(z), _p(q), _ld(q), _p(h), _ld(h), _p(dz), c, _p(dq), c,
              _p(dh), c, b * hh * ww, c, _stream())
    return dz, dq, dh


def upsample_flow_bwd(dout: Tensor, flow: Tensor, mask: Tensor):
    b, h, w, _ = flow.shape
    dflow = torch.zeros((b, h, w, 2), dtype=torch.float32, device=flow.device)
    dmask = empty_nhwc(b, h, w, 576, flow)
    _hip.call("ff_upsample_flow_bwd", _p(dout.contiguous()), _p(flow), _ld(flow), _p(mask), _ld(mask), _p(dflow), _p(dmask),
              b, h, w, _stream())
    return dflow, dmask
