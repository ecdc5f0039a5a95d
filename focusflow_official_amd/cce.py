"""Condition Control Encoder (FFE + CFE with fusion units) on the HIP path.

Mirrors the reference's module tree so that ``state_dict()`` keys and shapes
are interchangeable (parallel_fusion.py:153-274, extractor.py:6-56,118-192),
but the modules below only HOLD parameters: ``nn.Conv2d`` / ``nn.BatchNorm2d``
instances are never called.  The forward is a sequence of libfocusflow_hip
launches on NHWC fp32 tensors.
"""
from typing import Tuple, Optional, Sequence

import ctypes
import os
import weakref

import torch
import torch.nn as nn

from . import _hip, fn, ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID

EPS = 1e-5


_NORM_ON_LOAD = True      # ResidualBlock conv2 normalises while loading (inference; tests compare with the materialised route)


_ALL_PACKED = weakref.WeakSet()      # every PackedConv alive: prepack() finds the ones of a model here
_DMA_FRAG = True          # fragment-order weights for conv_dma.hip
_PREPACK = True           # all stale weight layouts of a training step in one launch (tests switch it off to compare)
_serial = [0]


class PackedConv:
    """Cache of one (or several Cout-concatenated) nn.Conv2d in kernel layout.

    Packed rows are [Cout][KH][KW][cin_pad]; re-packed by ff_pack_conv_weight
    whenever a source parameter changes (version counter) or moves.
    """
    # defaults for subclasses that build their own state (pwcnet._TrainPacked)
    cin_slices = None
    use_bias = True
    _w_raw = _wd_raw = _w_split = _wd_split = None
    _used_f = _used_d = False      # get() / get_dgrad() have been asked for: prepack() keeps these layouts fresh
    _gen = 0
    _frag = _frag_of = None
    _dfrag = _dfrag_of = None

    force_f32 = False

    def __init__(self, convs: Sequence[nn.Conv2d], cin_pad: Optional[int] = None,
                 cin_slices: Optional[Sequence[Tuple[int, int]]] = None, use_bias: bool = True, force_f32: bool = False):
        """cin_slices: take only these input-channel ranges of the weights, in this order (a convolution is linear
        in its input channels: the GRU splits off the part that meets the loop-invariant context features);
        use_bias=False leaves the bias to the other part (and its gradient: params() then
        holds None in the bias position)."""
        self.convs = list(convs)
        c0 = self.convs[0]
        self.kh, self.kw = c0.kernel_size
        self.stride = c0.stride[0]
        self.pad = tuple(c0.padding)
        self.dil = c0.dilation[0]
        self.cin_slices = list(cin_slices) if cin_slices is not None else None
        self.use_bias = use_bias
        self.force_f32 = force_f32      # exact-fp32 rows whatever the global conv precision (the range guard's repair route)
        self.cin = c0.in_channels if cin_slices is None else sum(hi - lo for lo, hi in cin_slices)
        self.cin_pad = cin_pad if cin_pad is not None else (self.cin + 3) // 4 * 4
        self.cout = sum(c.out_channels for c in self.convs)
        for c in self.convs:
            assert c.kernel_size == c0.kernel_size and c.in_channels == c0.in_channels and c.stride == c0.stride
        self._key = None
        self._gen = 0        # bumped by every repack of the forward rows: derived images (update_block's stage-major mask weights) key on it
        self.w = None
        self.b = None
        self._dkey = None
        self.wd = None
        self._w_raw = self._wd_raw = self._w_split = self._wd_split = None
        _serial[0] += 1
        self._serial = _serial[0]
        _ALL_PACKED.add(self)

    def _fwd_key(self):
        # (parameters through the module's own table: nn.Module.__getattr__ is several times slower, and a one-pair
        # forward asks for ~200 of these keys while it is host-bound)
        k = [ops.conv_precision()]
        for c in self.convs:
            w, b = c._parameters["weight"], c._parameters.get("bias")
            k += (w._version, w.data_ptr())
            if b is not None:
                k += (b._version, b.data_ptr())
        return tuple(k)

    def _dgrad_key(self):
        k = [ops.conv_precision()]
        for c in self.convs:
            w = c._parameters["weight"]
            k += (w._version, w.data_ptr())
        return tuple(k)

    def _small(self):
        # 1- and 2-channel 3x3 heads stay in fp32 rows: ff_conv2d_fwd runs them as dot products on the vector ALU
        # (conv_small.hip) instead of wasting a 64-wide matrix tile on them
        return self.cout <= 2 and (self.kh, self.kw, self.stride) == (3, 3, 1) and self.pad == (1, 1)

    def get(self):
        self._used_f = True
        key = self._fwd_key()
        if key != self._key:
            dev = self.convs[0].weight.device
            if self._w_raw is None or self._w_raw.device != dev:     # re-packed every training step: the buffers stay
                self._w_raw = torch.empty((self.cout, self.kh * self.kw * self.cin_pad), dtype=torch.float32, device=dev)
                self.b = torch.zeros(self.cout, dtype=torch.float32, device=dev)   # bias=False convs (SA) keep zeros
            self.w = self._w_raw
            off = 0
            for c in self.convs:
                wt = c.weight.detach()
                if self.cin_slices is not None:
                    wt = torch.cat([wt[:, lo:hi] for lo, hi in self.cin_slices], 1).contiguous()
                ops.pack_conv_weight(wt, self.w, self.cin_pad, off)
                if c.bias is not None and self.use_bias:
                    self.b[off:off + c.out_channels].copy_(c.bias.detach())  # device memcpy
                off += c.out_channels
            self.fmt = 0 if (self._small() or self.force_f32) else ops.w_format()
            if self.fmt != 0:
                self.w = ops.pack_split(self.w)
            self._key = key
            self._gen += 1
        return self.w, self.b

    def get_dgrad(self):
        """Weights of the input-gradient convolution: [cin_pad][KH][KW][cout_pad], flipped + transposed, in the
        active conv format (fp32 rows, or fp16-split rows: the dgrad then runs on the f16 matrix pipe with the
        gradient scaled by a power of two, see FFConvParams.x_amax).  Returns (rows, format)."""
        self._used_d = True
        key = self._dgrad_key()
        if key != self._dkey:
            cout_pad = (self.cout + 3) // 4 * 4
            dev = self.convs[0].weight.device
            if self._wd_raw is None or self._wd_raw.device != dev:   # zeroed once: the padding positions are never written
                self._wd_raw = torch.zeros((self.cin_pad, self.kh * self.kw * cout_pad), dtype=torch.float32, device=dev)
            self.wd = self._wd_raw
            off = 0
            for c in self.convs:
                wt = c.weight.detach()
                if self.cin_slices is not None:
                    wt = torch.cat([wt[:, lo:hi] for lo, hi in self.cin_slices], 1).contiguous()
                ops.pack_conv_weight_dgrad(wt, self.wd, cout_pad, off)
                off += c.out_channels
            self.dfmt = ops.w_format()
            if self.dfmt != 0:
                self.wd = ops.pack_split(self.wd)
            self._dkey = key
        return self.wd, self.dfmt

    def frag(self):
        """The forward rows in MFMA-fragment order (ops.pack_frag16) for the split-pair kernel; re-made per repack."""
        w, _ = self.get()
        key = (self._gen, self._key, w.data_ptr())      # (_key carries the parameters' versions: subclasses that repack into a fresh buffer never bump _gen)
        if self._frag_of != key:
            self._frag, self._frag_of = ops.pack_frag16(w, self.cout), key
        return self._frag

    def dma_f32_ok(self, xs) -> bool:
        """conv_dma.hip's fp32-input route takes this convolution (stride-1 'same' 3x3 / 1x5 / 5x1, f16x3, segments of
        multiples of 32 channels): it then wants the rows in fragment order too."""
        return (_DMA_FRAG and self.fmt == _hip.W_F16X3 and (self.kh, self.kw) in ((3, 3), (1, 5), (5, 1)) and self.stride == 1 and self.dil == 1
                and self.pad == (self.kh // 2, self.kw // 2) and all(x.shape[3] % 32 == 0 for x in xs))

    def frag_dgrad(self):
        """get_dgrad()'s rows in fragment order, or None where the input-gradient convolution does not take conv_dma.hip's
        fp32-input route (its 'input' channels are this convolution's padded output channels)."""
        wd, dfmt = self.get_dgrad()
        cout_pad = (self.cout + 3) // 4 * 4
        if not (_DMA_FRAG and dfmt == _hip.W_F16X3 and (self.kh, self.kw) in ((3, 3), (1, 5), (5, 1)) and self.stride == 1 and self.dil == 1
                and self.pad == (self.kh // 2, self.kw // 2) and cout_pad % 32 == 0):
            return None
        key = (self._dkey, wd.data_ptr())
        if self._dfrag_of != key:
            self._dfrag, self._dfrag_of = ops.pack_frag16(wd, self.cin_pad), key
        return self._dfrag

    def __call__(self, xs, act=ACT_NONE, **kw):
        w, b = self.get()
        if not isinstance(xs, (list, tuple)):
            xs = [xs]
        if _DMA_FRAG and self.fmt != 0 and (any(isinstance(x, ops.SplitT) for x in xs) or self.dma_f32_ok(xs)):     # conv_dma.hip loads its weights in fragment order
            kw["w_frag"] = self.frag()
        return ops.conv2d(xs, w, b, self.cout, self.kh, self.kw, self.stride, self.pad, act=act, w_fmt=self.fmt,
                          dilation=self.dil, **kw)

    def params(self):
        """Parameter tensors in the order ConvFn receives (and returns gradients for) them."""
        return [t for cv in self.convs for t in (cv.weight, cv.bias if self.use_bias else None)]

    def unpack_wgrad(self, dwp, j, off):
        """Packed weight gradient rows [off, off + Cout_j) -> gradient of convs[j].weight (zero outside cin_slices)."""
        cv = self.convs[j]
        part = ops.unpack_conv_wgrad(dwp, cv.out_channels, self.cin, self.kh, self.kw, self.cin_pad, off)
        if self.cin_slices is None:
            return part
        full = torch.zeros_like(cv.weight)
        o = 0
        for lo, hi in self.cin_slices:
            full[:, lo:hi] = part[:, o:o + hi - lo]
            o += hi - lo
        return full


def _pack_job(pc, need_f, need_d, dev):
    """FFPackJob of one PackedConv (its destination buffers are created here and stay): -> (job, fmt, dfmt)."""
    J = _hip.FFPackJob()
    convs = pc.convs
    off = 0
    for m, cv in enumerate(convs):
        J.w[m] = cv.weight.data_ptr()
        J.bias[m] = cv.bias.data_ptr() if (cv.bias is not None and pc.use_bias) else None
        J.cout_m[m], J.off[m] = cv.out_channels, off
        off += cv.out_channels
    J.nmem, J.cout, J.cin_src = len(convs), pc.cout, convs[0].in_channels
    sl = pc.cin_slices or []
    J.nslice = len(sl)
    for i, (lo, hi) in enumerate(sl):
        J.slice_lo[i], J.slice_hi[i] = lo, hi
    J.cin, J.cin_pad, J.KH, J.KW = pc.cin, pc.cin_pad, pc.kh, pc.kw
    fmt = 0 if (pc._small() or pc.force_f32) else ops.w_format()
    dfmt = ops.w_format()
    kf = pc.kh * pc.kw * pc.cin_pad
    if need_f:
        if pc.b is None or pc.b.device != dev or pc.b.shape[0] != pc.cout:
            pc.b = torch.zeros(pc.cout, dtype=torch.float32, device=dev)
        if fmt == 0:
            if pc._w_raw is None or pc._w_raw.device != dev:
                pc._w_raw = torch.empty((pc.cout, kf), dtype=torch.float32, device=dev)
            dst = pc._w_raw
        else:
            if pc._w_split is None or pc._w_split.device != dev:
                pc._w_split = torch.empty((pc.cout, (kf + 31) // 32 * 128), dtype=torch.uint8, device=dev)
            dst = pc._w_split
        J.fwd, J.fwd_format = dst.data_ptr(), fmt
        J.bias_dst = pc.b.data_ptr() if pc.use_bias else None
        J.items_fwd = pc.cout * ((kf + 31) // 32) * 4
    cout_pad = (pc.cout + 3) // 4 * 4
    kd = pc.kh * pc.kw * cout_pad
    if need_d:
        if dfmt == 0:
            if pc._wd_raw is None or pc._wd_raw.device != dev:
                pc._wd_raw = torch.zeros((pc.cin_pad, kd), dtype=torch.float32, device=dev)
            dst = pc._wd_raw
        else:
            if pc._wd_split is None or pc._wd_split.device != dev:
                pc._wd_split = torch.empty((pc.cin_pad, (kd + 31) // 32 * 128), dtype=torch.uint8, device=dev)
            dst = pc._wd_split
        J.dgrad, J.dgrad_format, J.cout_pad = dst.data_ptr(), dfmt, cout_pad
        J.items_dgrad = pc.cin_pad * ((kd + 31) // 32) * 4
    return J, fmt, dfmt


def prepack(owner: nn.Module, device) -> int:
    """Bring every stale weight layout of `owner`'s convolutions up to date in ONE launch (ff_pack_weights_table) - the
    layouts this process has asked for before (PackedConv.get / get_dgrad mark them), i.e. from the second training step
    on: the optimiser has just changed every parameter, and packing lazily costs ~7 small launches per convolution per
    step.  The job table lives on the device and is rebuilt only when a pointer, a format or the set of stale layouts
    changes.  Returns the number of PackedConvs packed (0: nothing stale, or FF_PREPACK=0)."""
    if not _PREPACK:
        return 0
    ids = owner.__dict__.get("_ff_param_ids")
    if ids is None or ids[0] != sum(1 for _ in owner.parameters()):
        ids = owner.__dict__["_ff_param_ids"] = (sum(1 for _ in owner.parameters()), {id(p) for p in owner.parameters()})
    device = torch.device(device)
    # the owner's own PackedConvs, found once per (parameter set, PackedConvs ever made): a process with several models
    # alive (bench.py's parity legs) otherwise pays for scanning and sorting all of theirs on every training step
    mine = owner.__dict__.get("_ff_packed_mine")
    if mine is None or mine[0] != (ids[0], _serial[0]):
        found = sorted((pc for pc in _ALL_PACKED if type(pc) is PackedConv and id(pc.convs[0].weight) in ids[1]),
                       key=lambda pc: pc._serial)
        mine = owner.__dict__["_ff_packed_mine"] = ((ids[0], _serial[0]), [weakref.ref(pc) for pc in found])
    todo = []
    for ref in mine[1]:
        pc = ref()
        if pc is None or not (pc._used_f or pc._used_d):
            continue
        w0 = pc.convs[0].weight
        if id(w0) not in ids[1] or w0.device != device or len(pc.convs) > _hip.PACK_MAX_MEMBERS \
                or len(pc.cin_slices or ()) > _hip.PACK_MAX_SLICES:
            continue
        kf = pc._fwd_key() if pc._used_f else None
        kd = pc._dgrad_key() if pc._used_d else None
        nf, nd = pc._used_f and kf != pc._key, pc._used_d and kd != pc._dkey
        if nf or nd:
            todo.append((pc, nf, nd, kf, kd))
    if len(todo) < 4:           # a few stragglers: get() / get_dgrad() pack them on demand
        return 0
    jobs = [_pack_job(pc, nf, nd, device) for pc, nf, nd, _, _ in todo]
    sig = tuple((pc._serial, nf, nd, bytes(J)) for (pc, nf, nd, _, _), (J, _, _) in zip(todo, jobs))
    cache = owner.__dict__.get("_ff_pack_table")
    if cache is None or cache[0] != sig:
        arr = (_hip.FFPackJob * len(jobs))()
        blk = 0
        for i, (J, _, _) in enumerate(jobs):
            J.block0 = blk
            _hip.call("ff_pack_job_check", ctypes.byref(J))
            blk += (J.items_fwd + J.items_dgrad + 255) // 256
            arr[i] = J
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        cache = owner.__dict__["_ff_pack_table"] = (sig, host.to(device), len(jobs), blk)
    ops.pack_weights_table(cache[1], cache[2], cache[3])
    for (pc, nf, nd, kf, kd), (J, fmt, dfmt) in zip(todo, jobs):
        if nf:
            pc.w, pc.fmt, pc._key = (pc._w_raw if fmt == 0 else pc._w_split), fmt, kf
            pc._gen += 1
        if nd:
            pc.wd, pc.dfmt, pc._dkey = (pc._wd_raw if dfmt == 0 else pc._wd_split), dfmt, kd
    return len(todo)


def train_streams() -> bool:
    """The encoder / branch streams in RECORDED (training) passes: autograd replays every backward node on the stream its
    forward ran on.  FF_TRAIN_STREAMS=1 / 0 decides; unset, they are on for single-process training and OFF once a
    torch.distributed process group exists: DDP's bucket hooks would then fire from the side streams, a choreography no
    multi-rank RCCL run has exercised yet (the 2-rank DDP test runs both ways on one card)."""
    v = os.environ.get("FF_TRAIN_STREAMS")
    if v is not None:
        return v != "0"
    return not (torch.distributed.is_available() and torch.distributed.is_initialized())
_BRANCH_STREAMS = True    # mask branch of the CCE encoder on a side stream
_branch_streams = {}


def _branch_stream(device):
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)     # one per caller stream
    if key not in _branch_streams:
        _branch_streams[key] = torch.cuda.Stream(device=device)
    return _branch_streams[key]


_FUSION_PAIR = os.environ.get("FF_FUSION_PAIR", "1") != "0"   # fusion units 1-3 as ff_fusion_pair_fwd launches with lazy normalised inputs (inference; A/B switch)
_PAIR_FUSION = True       # both 1x1 convs of a fusion unit in one launch where ff_fusion_pair_fwd has no instance (inference)


def invalidate_packed(module: nn.Module) -> int:
    """Drop every cached derived tensor under `module`: packed / split conv weights (PackedConv) and folded BatchNorm
    coefficients (ops.bn_fold).  The caches are keyed on the parameters' version counters and storage, which an in-place
    write through `.data` (EMA weight swaps, `weight.data.clamp_()`, legacy optimisers) does not change - call this
    after such an update.  `load_state_dict` and `train()` / `eval()` of the top-level modules call it themselves.
    Returns the number of caches dropped."""
    n = 0
    for m in module.modules():
        for v in vars(m).values():
            vs = v if isinstance(v, (list, tuple)) else (v,)
            for pc in vs:
                if isinstance(pc, PackedConv):
                    pc._key = pc._dkey = None
                    n += 1
        if hasattr(m, "_ff_fold"):
            del m._ff_fold
            n += 1
        if getattr(m, "_pair_key", None) is not None:
            m._pair_key = None
            n += 1
        if getattr(m, "_mask2_stage_of", None) is not None:     # BasicUpdateBlock.upsample's stage-major image of mask[2]
            m._mask2_stage = m._mask2_stage_of = None
            n += 1
    return n


def _make_norm(kind: str, c: int):
    if kind == "batch":
        return nn.BatchNorm2d(c)
    if kind == "instance":
        return nn.InstanceNorm2d(c)  # no parameters, no buffers: contributes no state_dict keys
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"norm_fn {kind} is not supported on the HIP path")


class ResidualBlock(nn.Module):
    """Parameter holder with the reference's names (extractor.py:6-46)."""

    def __init__(self, in_planes, planes, norm_fn, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.norm1 = _make_norm(norm_fn, planes)
        self.norm2 = _make_norm(norm_fn, planes)
        self.stride = stride
        self.downsample = None
        if stride != 1:
            # the reference registers this norm twice: as norm3 and as downsample.1
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)
        self._p1, self._p2 = PackedConv([self.conv1]), PackedConv([self.conv2])
        self._pd = PackedConv([self.downsample[0]]) if stride != 1 else None


class _FusionConv(nn.Module):
    """Conv1x1 / Concat fusion operators (parallel_fusion.py:76-95): `.conv` only."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)
        self._p = PackedConv([self.conv])


class _SA(nn.Module):
    """Spatial-attention fusion (parallel_fusion.py:49-73): out = sigmoid(conv3x3([mean_c q1, max_c q1])) * conv_v(v) + q,
    q1 = conv_q(cat[q, v]); every conv is 3x3 without bias.  Parameter names as in the reference."""

    def __init__(self, c, bias=False):
        super().__init__()
        self.conv_q = nn.Conv2d(2 * c, c, 3, padding=1, bias=bias)
        self.conv_v = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=bias))
        self.s_map = nn.Sequential(nn.Conv2d(2, 1, 3, 1, 1, bias=bias), nn.Sigmoid())
        self._pq, self._pv, self._ps = PackedConv([self.conv_q]), PackedConv([self.conv_v[0]]), PackedConv([self.s_map[0]])

    def run(self, q, v):
        q1 = fn.conv(self._pq, [q, v])
        v1 = fn.conv(self._pv, v)
        s = fn.conv(self._ps, fn.chan_stats(q1), act=ACT_SIGMOID)          # (B,H,W,1)
        return fn.scale_add(v1, s, q, 0)


class _CA(nn.Module):
    """Channel-attention fusion (parallel_fusion.py:14-46): out = (mlp(avg_hw q1) + mlp(max_hw q1)) * conv_v(v) + q,
    mlp = conv1x1(C -> C/16) -> ReLU -> conv1x1(-> C) -> Sigmoid.  The two pooled vectors run through the MLP as
    one batch of 2B one-pixel images."""

    def __init__(self, c, reduction=16, bias=True):
        super().__init__()
        self.conv_q = nn.Conv2d(2 * c, c, 3, padding=1, bias=bias)
        self.conv_v = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=bias))
        self.avgpool, self.maxpool = nn.AdaptiveAvgPool2d(1), nn.AdaptiveMaxPool2d(1)   # no state; kept for module parity
        self.c_map = nn.Sequential(nn.Conv2d(c, c // reduction, 1, padding=0, bias=bias), nn.ReLU(inplace=True),
                                   nn.Conv2d(c // reduction, c, 1, padding=0, bias=bias), nn.Sigmoid())
        self._pq, self._pv = PackedConv([self.conv_q]), PackedConv([self.conv_v[0]])
        self._p1, self._p2 = PackedConv([self.c_map[0]]), PackedConv([self.c_map[2]])

    def run(self, q, v):
        q1 = fn.conv(self._pq, [q, v])
        v1 = fn.conv(self._pv, v)
        pooled = fn.spatial_stats(q1)                                       # (2B,1,1,C): avg rows, then max rows
        ch = self._p1.cout                                                  # C/16 channels, padded to a multiple of 4
        hid = fn.conv(self._p1, pooled, act=ACT_RELU, pad_out=True,
                      fill_tail=(lambda full: full[..., ch:].zero_()) if ch % 4 else None)
        cm = fn.conv(self._p2, hid, act=ACT_SIGMOID)                        # (2B,1,1,C)
        return fn.scale_add(v1, cm, q, 1)


class FusionUnit(nn.Module):
    """parallel_fusion.py:98-150: '1x1conv', '1x1conv-unidirection', 'concat', 'SA', 'CA'."""

    def __init__(self, c, fusion_type, bi_direction=True):
        super().__init__()
        self.fusion_type = fusion_type
        if fusion_type in ("1x1conv", "1x1conv-unidirection"):
            self.mask2img = _FusionConv(c, c)
            self.img2mask = _FusionConv(c, c) if (bi_direction and fusion_type == "1x1conv") else None
        elif fusion_type == "concat":
            self.mask2img = _FusionConv(2 * c, c)
            self.img2mask = _FusionConv(2 * c, c) if bi_direction else None
        elif fusion_type in ("SA", "CA"):
            unit = _SA if fusion_type == "SA" else _CA
            self.mask2img = unit(c)
            self.img2mask = unit(c) if bi_direction else None
        else:
            raise ValueError(f"Fusion type {fusion_type} not supported.")

    def _paired(self):
        """Both 1x1 convs of the unit as one packed weight over the input segments [img, mask] (2C channels):
        rows 0..C-1 (img') read the mask channels, rows C..2C-1 (mask') the img channels - an anti-diagonal block
        matrix.  Rebuilt when a parameter changes (same cache key as PackedConv)."""
        a, b = self.mask2img.conv, self.img2mask.conv
        key = (ops.conv_precision(),) + tuple((t._version, t.data_ptr()) for t in (a.weight, a.bias, b.weight, b.bias))
        if getattr(self, "_pair_key", None) != key:
            c = a.out_channels
            wf = torch.zeros((2 * c, 2 * c, 1, 1), dtype=torch.float32, device=a.weight.device)
            wf[:c, c:] = a.weight.detach()       # img' <- mask
            wf[c:, :c] = b.weight.detach()       # mask' <- img
            rows = torch.empty((2 * c, 2 * c), dtype=torch.float32, device=wf.device)
            ops.pack_conv_weight(wf, rows, 2 * c, 0)
            self._pair_w = ops.pack_split(rows)
            self._pair_b = torch.cat([a.bias.detach(), b.bias.detach()]).contiguous()
            self._pair_key = key
        return self._pair_w, self._pair_b

    def _pair_kernel_ok(self, img):
        """ff_fusion_pair_fwd takes this unit: both directions, inference, a split weight format, 64 or 96 channels per
        branch, whole tiles per image."""
        if not (_FUSION_PAIR and self.fusion_type == "1x1conv" and self.img2mask is not None and not torch.is_grad_enabled()
                and ops.w_format() in (_hip.W_F16X3, _hip.W_F16)):
            return False
        b, h, w, c = img.shape
        tp = ops.fusion_pair_tile(c)
        return tp > 0 and (h * w) % tp == 0

    def run(self, mask, img):
        if self._pair_kernel_ok(img) and mask.shape == img.shape and ops.fusion_pair_inputs_ok(img, mask):
            # inference: the unit as ONE bandwidth-shaped launch (csrc/fusion_pair.hip); lazy inputs (ops.LazyAct: the
            # normalisation pass in front of the unit) are evaluated inside its loader
            pa, pb = self.mask2img._p, self.img2mask._p
            (_, ba), (_, bb) = pa.get(), pb.get()
            img_out, mask_out = ops.fusion_pair(img, mask, (pa.frag(), pb.frag()), (ba, bb), ops.w_format())
            return mask_out, img_out
        if isinstance(img, ops.LazyAct):
            img = img.materialise()
        if isinstance(mask, ops.LazyAct):
            mask = mask.materialise()
        if (_PAIR_FUSION and self.fusion_type == "1x1conv" and self.img2mask is not None and not torch.is_grad_enabled()
                and ops.w_format() != 0 and img.shape[3] % 32 == 0 and img.shape == mask.shape):
            # inference: ONE launch for img' = img + conv(mask) and mask' = mask + conv(img).  The two separate launches
            # read both tensors twice (once as input, once as residual) and are HBM-bound at 192x256; here every tensor
            # is read once from HBM (the residual read hits L2) at the price of the zero blocks' MFMAs, which were idle.
            c = img.shape[3]
            w, bias = self._paired()
            y = ops.conv2d([img, mask], w, bias, 2 * c, 1, 1, res=img, res2=mask, res_split=c, w_fmt=ops.w_format())
            return y[..., c:], y[..., :c]
        if self.fusion_type in ("SA", "CA"):
            img_out = self.mask2img.run(img, mask)
            mask_out = self.img2mask.run(mask, img) if self.img2mask is not None else mask
        elif self.fusion_type == "concat":
            img_out = fn.conv(self.mask2img._p, [img, mask])
            mask_out = fn.conv(self.img2mask._p, [mask, img]) if self.img2mask is not None else mask
        else:  # out = q + conv1x1(v): the residual add rides in the conv epilogue
            img_out = fn.conv(self.mask2img._p, mask, res=img)
            mask_out = fn.conv(self.img2mask._p, img, res=mask) if self.img2mask is not None else mask
        return mask_out, img_out


class BasicParallelFusionLayer(nn.Module):
    """CCE = frame branch (FFE) + condition branch (CFE) + 5 fusion units.

    Same constructor and attribute names as parallel_fusion.py:153-209; call
    with NHWC4 tensors from ops.prep_input, returns NHWC (B, H/8, W/8, output_dim).
    """

    def __init__(self, img_channel=3, mask_channel=3, output_dim=128, norm_fn="batch", dropout=0, cfg=None):
        super().__init__()
        self.norm_fn = norm_fn
        self.fusion_type = cfg.MODEL.FUSION_TYPE
        if dropout and dropout > 0:
            raise NotImplementedError("dropout > 0 is not used by any reference config and is not built")
        self.norm1 = _make_norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(img_channel, 64, 7, stride=2, padding=3)
        self.layer1 = self._stage(64, 64, 1)
        self.layer2 = self._stage(64, 96, 2)
        self.layer3 = self._stage(96, 128, 2)
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.mask_norm1 = _make_norm(norm_fn, 64)
        self.mask_conv1 = nn.Conv2d(mask_channel, 64, 7, stride=2, padding=3)
        self.fusion1 = FusionUnit(64, self.fusion_type, True)
        self.fusion2 = FusionUnit(64, self.fusion_type, True)
        self.fusion3 = FusionUnit(96, self.fusion_type, True)
        self.fusion4 = FusionUnit(128, self.fusion_type, True)
        self.fusion5 = FusionUnit(output_dim, self.fusion_type, False)
        self.mask_layer1 = self._stage(64, 64, 1)
        self.mask_layer2 = self._stage(64, 96, 2)
        self.mask_layer3 = self._stage(96, 128, 2)
        self.mask_conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = None
        # parallel_fusion.py:198-205 initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._stem = PackedConv([self.conv1], 4)
        self._mstem = PackedConv([self.mask_conv1], 4)
        self._out = PackedConv([self.conv2])
        self._mout = PackedConv([self.mask_conv2])

    def _stage(self, cin, cout, stride):
        return nn.Sequential(ResidualBlock(cin, cout, self.norm_fn, stride), ResidualBlock(cout, cout, self.norm_fn, 1))

    # -- execution ---------------------------------------------------------
    def _conv_norm(self, x, pc: PackedConv, norm, act, res=None, lazy=False, take_res=False, defer_res=False):
        """act(norm(conv(x))) and, with `res`, relu(res + that).  lazy (InstanceNorm inference only): -> ops.LazyAct, the
        normalisation left to the fusion unit that reads it.  take_res / defer_res (recorded passes): fn.res_grad_fused."""
        relu = act == ACT_RELU
        params = [pc.convs[0].weight, pc.convs[0].bias]
        if self.norm_fn == "batch":
            params += [norm.weight, norm.bias]
        taped = fn.recording(x, res, *params)
        if self.norm_fn == "instance":
            if not taped:
                y, st = pc(x, want_stats=True)       # the statistics come out of the conv's epilogue where it can
                if lazy:
                    return ops.LazyAct(y, st, y.shape[1] * y.shape[2], EPS, act, res)
                return ops.norm_apply(y, st, True, EPS, act=act, res=res, out=y)
            if ops.CONV_STATS_TRAIN:
                y, st = fn.conv(pc, x, want_stats=True, take_res=take_res)
            else:
                y = fn.conv(pc, x, take_res=take_res)
                st = ops.norm_stats(y.detach(), per_sample=True)
            return fn.NormFn.apply(y, None, None, res, True, False, EPS, relu, st, defer_res)
        if self.norm_fn == "batch":
            if norm.training:
                y = fn.conv(pc, x, take_res=take_res and taped)
                st = ops.norm_stats(y.detach(), per_sample=False)
                b, h, w, _ = y.shape
                ops.bn_update_running(norm, st, b * h * w)
                norm.num_batches_tracked += 1
                if taped:
                    return fn.NormFn.apply(y, norm.weight, norm.bias, res, False, False, norm.eps, relu, st, defer_res)
                return ops.norm_apply(y, st, False, norm.eps, norm.weight, norm.bias, act=act, res=res, out=y)
            if taped:  # frozen BatchNorm inside a training step (raft.py:104-107): fixed statistics
                y = fn.conv(pc, x, take_res=take_res)
                n = float(y.shape[0] * y.shape[1] * y.shape[2])
                rm, rv = norm.running_mean.double(), norm.running_var.double()
                st = torch.stack([rm * n, (rv + rm * rm) * n], -1)[None].contiguous()
                return fn.NormFn.apply(y, norm.weight, norm.bias, res, False, True, norm.eps, relu, st, defer_res)
            sc, sh = ops.bn_fold(norm)  # eval: scale/shift ride in the conv epilogue
            return pc(x, act=act, ch_scale=sc, ch_shift=sh, res=res, act_res=ACT_RELU)
        if taped:
            raise NotImplementedError("norm_fn='none' has no autograd path (unused by the reference configs)")
        return pc(x, act=act, res=res, act_res=ACT_RELU)  # 'none'

    def _block_recorded(self, blk: ResidualBlock, x) -> bool:
        """Both convolutions of the block and its two norms go through the tape, with x itself differentiated."""
        return torch.is_grad_enabled() and x.requires_grad and self.norm_fn in ("instance", "batch")

    def _block(self, blk: ResidualBlock, x, lazy_out=False):
        p2 = blk._p2
        if (_NORM_ON_LOAD and self.norm_fn == "instance" and not torch.is_grad_enabled() and ops.w_format() in (_hip.W_F16X3, _hip.W_F16)
                and p2.cin % 32 == 0 and (p2.kh, p2.kw, p2.stride) == (3, 3, 1)):
            # inference: conv2 applies relu(norm1(.)) while it loads conv1's raw output - one full read + write of the
            # activation less per block (same coefficients, same arithmetic as ff_norm_apply: bit-identical results)
            t1, st = blk._p1(x, want_stats=True)
            sc, sh = ops.norm_coeffs(st, t1.shape[1] * t1.shape[2], EPS)
            if blk.downsample is not None:
                x = self._conv_norm(x, blk._pd, blk.norm3, ACT_NONE)
            t2, st2 = p2(t1, in_scale=sc, in_shift=sh, in_act=ACT_RELU, want_stats=True)
            if lazy_out:     # the stage's last block: relu(x + relu(norm2(t2))) is evaluated by the fusion unit's loader
                return ops.LazyAct(t2, st2, t2.shape[1] * t2.shape[2], EPS, ACT_RELU, x)
            return ops.norm_apply(t2, st2, True, EPS, act=ACT_RELU, res=x, out=t2)
        # recorded passes, plain blocks (the input is the skip connection): the skip's gradient is added inside conv1's
        # input-gradient launch instead of by an autograd add over two activation-sized tensors
        fuse = blk.downsample is None and self._block_recorded(blk, x) and fn.res_grad_fused(blk._p1)
        y = self._conv_norm(x, blk._p1, blk.norm1, ACT_RELU, take_res=fuse)
        if blk.downsample is not None:
            x = self._conv_norm(x, blk._pd, blk.norm3, ACT_NONE)
        return self._conv_norm(y, blk._p2, blk.norm2, ACT_RELU, res=x, lazy=lazy_out and self.norm_fn == "instance" and not torch.is_grad_enabled(),
                               defer_res=fuse)

    def _run_stage(self, stage, x, lazy_out=False):
        if lazy_out:
            return self._block(stage[1], self._block(stage[0], x), True)
        return self._block(stage[1], self._block(stage[0], x))

    def _branches(self, fm, fx, m, x):
        """(fm(m), fx(x)): the mask branch and the image branch between two fusion units are independent.  Inference runs
        the mask branch on a side stream (forked and joined with events: capturable): one branch's memory-bound norm passes
        overlap the other's convolutions."""
        # (not while a hipGraph is being captured: the capture takes ONE level of forks - the context encoder beside the feature
        # encoder, raft_net.py, replay 665 against 650 pairs/s without it - but a fork inside a forked stream dumps core in the HIP
        # runtime of ROCm 7.2: measured, round 4)
        if not (_BRANCH_STREAMS and ops.policy.encoder_streams_ok and not ops.policy.single_stream and (not torch.is_grad_enabled() or train_streams()) and x.is_cuda) or torch.cuda.is_current_stream_capturing():
            return fm(m), fx(x)
        main = torch.cuda.current_stream()
        side = _branch_stream(x.device)
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            mo = fm(m)
            join = torch.cuda.Event()
            join.record(side)
        m.record_stream(side)              # allocated on the main stream, read on the side stream
        xo = fx(x)
        main.wait_event(join)
        mo.record_stream(main)             # allocated on the side stream, read on the main stream
        return mo, xo

    def _lazy_for(self, unit: "FusionUnit", c, h, w):
        """Leave the normalisation pass in front of `unit` to the unit's own loader (ops.LazyAct)?  InstanceNorm inference
        only - an eval BatchNorm is already folded into the producing convolution's epilogue."""
        if self.norm_fn != "instance" or torch.is_grad_enabled():
            return False
        probe = torch.empty((1, h, w, c), device="meta")
        return unit._pair_kernel_ok(probe)

    def forward(self, x, mask):
        b, hh, ww = x.shape[0], x.shape[1], x.shape[2]
        lz1 = self._lazy_for(self.fusion1, 64, hh // 2, ww // 2)
        lz2 = self._lazy_for(self.fusion2, 64, hh // 2, ww // 2)
        lz3 = self._lazy_for(self.fusion3, 96, hh // 4, ww // 4)
        m, x = self._branches(lambda t: self._conv_norm(t, self._mstem, self.mask_norm1, ACT_RELU, lazy=lz1),
                              lambda t: self._conv_norm(t, self._stem, self.norm1, ACT_RELU, lazy=lz1), mask, x)
        m, x = self.fusion1.run(m, x)
        m, x = self._branches(lambda t: self._run_stage(self.mask_layer1, t, lz2), lambda t: self._run_stage(self.layer1, t, lz2), m, x)
        m, x = self.fusion2.run(m, x)
        m, x = self._branches(lambda t: self._run_stage(self.mask_layer2, t, lz3), lambda t: self._run_stage(self.layer2, t, lz3), m, x)
        m, x = self.fusion3.run(m, x)
        m, x = self._branches(lambda t: self._run_stage(self.mask_layer3, t), lambda t: self._run_stage(self.layer3, t), m, x)
        m, x = self.fusion4.run(m, x)
        m, x = fn.conv(self._mout, m), fn.conv(self._out, x)
        m, x = self.fusion5.run(m, x)
        return x

    # -- reference API -----------------------------------------------------
    def freeze_self(self, mode):
        if mode == "parallel":  # parallel_fusion.py:249-267: freeze the frame branch
            for mod in (self.conv1, self.norm1, self.layer1, self.layer2, self.layer3, self.conv2):
                for p in mod.parameters():
                    p.requires_grad = False

    def copy_to_branch(self):  # parallel_fusion.py:269-274
        self.mask_conv1.load_state_dict(self.conv1.state_dict())
        self.mask_layer1.load_state_dict(self.layer1.state_dict())
        self.mask_layer2.load_state_dict(self.layer2.state_dict())
        self.mask_layer3.load_state_dict(self.layer3.state_dict())
        self.mask_conv2.load_state_dict(self.conv2.state_dict())
