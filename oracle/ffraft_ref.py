"""CPU oracle for the FF-RAFT hot path — TEST INFRASTRUCTURE, not product code.

A functional (state_dict-driven) plain-PyTorch fp32 restatement of the
reference's FF-RAFT forward.  Parameters are looked up by the reference's own
state_dict keys, so the reference, this oracle and the HIP model can share
weights.  Parity status: PINNED by ``tests/golden/*.npz`` (generated from the
reference's FF_RAFT_Core modules by ``tests/golden/make_golden.py``).

Reference lines followed (relative to core/models/ff-raft/FF_RAFT_Core/):
  wrapper        ff_raft.py:31-38 (point masks), :134-160 (scaling + dispatch)
  RAFT graph     raft.py:173-236, upsample_flow raft.py:159-170
  CCE encoder    parallel_fusion.py:211-247, FusionUnit :142-150,
                 Conv1x1 :87-95, Concat :76-84
  residual block extractor.py:48-56
  CorrBlock      corr.py:13-27 (pyramid), :29-50 (lookup), :52-60 (volume)
  sampler        utils/utils.py:57-71, coords_grid :74-77
  update block   update.py:89-97, :45-60, :13-14, :126-135
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# layer helpers
# ----------------------------------------------------------------------------
# None, or a callable applied to BOTH operands of every convolution and of the correlation matmul: lets the tests measure
# what the reference's own GPU arithmetic does to the flow (common.py:25-27 turns TF32 on for cudnn and matmul; every
# shipped config sets ALLOW_TF32: true) - see round_tf32() and tests/golden/make_golden_tf32.py.
OPERAND_ROUND = None


def round_tf32(t: Tensor) -> Tensor:
    """fp32 -> TF32 (10 explicit mantissa bits, round to nearest even) -> fp32, as the tensor cores read their operands."""
    if t.dtype != torch.float32:
        return t
    i = t.contiguous().view(torch.int32)
    r = (i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF
    return r.view(torch.float32)


def _conv(sd, name, x, stride=1, padding=0):
    w = sd[name + ".weight"]
    if OPERAND_ROUND is not None:
        x, w = OPERAND_ROUND(x), OPERAND_ROUND(w)
    return F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=padding)   # SA convs: no bias


def _norm(sd, name, x, kind, training):
    if kind == "instance":  # nn.InstanceNorm2d default: no affine, no running stats
        return F.instance_norm(x, eps=EPS)
    if kind == "batch":
        return F.batch_norm(
            x, sd[name + ".running_mean"], sd[name + ".running_var"],
            sd[name + ".weight"], sd[name + ".bias"],
            training=training, momentum=BN_MOMENTUM, eps=EPS)
    if kind == "none":
        return x
    raise ValueError(kind)


def _resblock(sd, p, x, kind, stride, training):
    """extractor.py:48-56.  The stride-2 block's downsample norm is the module
    registered as both ``norm3`` and ``downsample.1``; ``downsample.1`` is the
    key that wins on load, so that is the one read here."""
    y = torch.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, stride, 1), kind, training))
    y = torch.relu(_norm(sd, p + ".norm2", _conv(sd, p + ".conv2", y, 1, 1), kind, training))
    if stride != 1:
        x = _norm(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), kind, training)
    return torch.relu(x + y)


def _stage(sd, p, x, kind, stride, training):
    x = _resblock(sd, p + ".0", x, kind, stride, training)
    return _resblock(sd, p + ".1", x, kind, 1, training)


def _fuse(sd, p, mask, img, fusion_type, bidirectional):
    """parallel_fusion.py:142-150 — both outputs come from the PRE-fusion pair."""
    if fusion_type in ("1x1conv", "1x1conv-unidirection"):
        img_out = img + _conv(sd, p + ".mask2img.conv", mask)
        if bidirectional and fusion_type == "1x1conv":
            mask_out = mask + _conv(sd, p + ".img2mask.conv", img)
        else:
            mask_out = mask
    elif fusion_type == "concat":
        img_out = _conv(sd, p + ".mask2img.conv", torch.cat([img, mask], 1))
        mask_out = _conv(sd, p + ".img2mask.conv", torch.cat([mask, img], 1)) if bidirectional else mask
    elif fusion_type in ("SA", "CA"):
        unit = _sa_unit if fusion_type == "SA" else _ca_unit
        img_out = unit(sd, p + ".mask2img", img, mask)
        mask_out = unit(sd, p + ".img2mask", mask, img) if bidirectional else mask
    else:
        raise ValueError(f"Fusion type {fusion_type} not supported.")
    return mask_out, img_out


def _sa_unit(sd, p, q, v):
    """SA.forward (parallel_fusion.py:63-73): spatial map from the channel mean / max of conv_q(cat[q, v])."""
    q1 = _conv(sd, p + ".conv_q", torch.cat([q, v], 1), padding=1)
    v = _conv(sd, p + ".conv_v.0", v, padding=1)
    st = torch.cat([q1.mean(dim=1, keepdim=True), q1.max(dim=1, keepdim=True)[0]], 1)
    return torch.sigmoid(_conv(sd, p + ".s_map.0", st, padding=1)) * v + q


def _ca_unit(sd, p, q, v):
    """CA.forward (parallel_fusion.py:39-46): channel map = mlp(avg-pooled) + mlp(max-pooled) of conv_q(cat[q, v])."""
    q1 = _conv(sd, p + ".conv_q", torch.cat([q, v], 1), padding=1)
    v = _conv(sd, p + ".conv_v.0", v, padding=1)

    def mlp(t):
        return torch.sigmoid(_conv(sd, p + ".c_map.2", torch.relu(_conv(sd, p + ".c_map.0", t))))

    c_map = mlp(q1.mean(dim=(2, 3), keepdim=True)) + mlp(q1.amax(dim=(2, 3), keepdim=True))
    return c_map * v + q


def cce_encoder(sd, p, x, mask, kind, training=False, fusion_type="1x1conv"):
    """Condition Control Encoder, parallel_fusion.py:211-247."""
    m = torch.relu(_norm(sd, p + ".mask_norm1", _conv(sd, p + ".mask_conv1", mask, 2, 3), kind, training))
    x = torch.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, 2, 3), kind, training))
    m, x = _fuse(sd, p + ".fusion1", m, x, fusion_type, True)
    m = _stage(sd, p + ".mask_layer1", m, kind, 1, training)
    x = _stage(sd, p + ".layer1", x, kind, 1, training)
    m, x = _fuse(sd, p + ".fusion2", m, x, fusion_type, True)
    m = _stage(sd, p + ".mask_layer2", m, kind, 2, training)
    x = _stage(sd, p + ".layer2", x, kind, 2, training)
    m, x = _fuse(sd, p + ".fusion3", m, x, fusion_type, True)
    m = _stage(sd, p + ".mask_layer3", m, kind, 2, training)
    x = _stage(sd, p + ".layer3", x, kind, 2, training)
    m, x = _fuse(sd, p + ".fusion4", m, x, fusion_type, True)
    m = _conv(sd, p + ".mask_conv2", m)
    x = _conv(sd, p + ".conv2", x)
    m, x = _fuse(sd, p + ".fusion5", m, x, fusion_type, False)
    return x


# ----------------------------------------------------------------------------
# correlation volume, pyramid, lookup
# ----------------------------------------------------------------------------
def corr_volume(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """corr.py:52-60 → (B, Q, H, W) with Q = H*W, scaled by 1/sqrt(C)."""
    b, c, h, w = fmap1.shape
    if OPERAND_ROUND is not None:
        fmap1, fmap2 = OPERAND_ROUND(fmap1), OPERAND_ROUND(fmap2)
    vol = torch.matmul(fmap1.view(b, c, h * w).transpose(1, 2), fmap2.view(b, c, h * w))
    return (vol / torch.sqrt(torch.tensor(c).float()).to(vol.dtype)).view(b, h * w, h, w)


def corr_pyramid(vol: Tensor, num_levels=4, half: bool = False) -> List[Tensor]:
    """corr.py:21-27 → list of (B*Q, 1, h_l, w_l).

    half=True is the fp16 correlation pyramid of BASELINE configs[4] (the storage torch autocast would give CorrBlock:
    the matmul result and every avg_pool2d output are rounded to fp16, each pooling reads the ROUNDED level below and
    sums in fp32; grid_sample is on autocast's fp32 list, so the lookup interpolates in fp32 on the fp16 values).
    Returned as fp32 tensors that hold fp16-representable values."""
    b, q, h, w = vol.shape
    lvl = vol.reshape(b * q, 1, h, w)
    if half:
        lvl = lvl.half().float()
    out = [lvl]
    for _ in range(num_levels - 1):
        lvl = F.avg_pool2d(lvl, 2, stride=2)
        if half:
            lvl = lvl.half().float()
        out.append(lvl)
    return out


def coords_grid(b, h, w, dtype=torch.float32) -> Tensor:
    """utils.py:74-77 — channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], 0).to(dtype)[None].repeat(b, 1, 1, 1)


def corr_lookup(pyramid: List[Tensor], coords: Tensor, radius=4) -> Tensor:
    """corr.py:29-50 via grid_sample (the reference's own route).

    Window axis 0 carries the x-offset (the reference's dx/dy naming is
    swapped): channel k = level*81 + a*9 + b, a = x-offset idx, b = y-offset idx.
    """
    b, _, h, w = coords.shape
    r = radius
    pts = coords.permute(0, 2, 3, 1).reshape(b * h * w, 1, 1, 2)
    off = torch.linspace(-r, r, 2 * r + 1, dtype=coords.dtype)
    delta = torch.stack(torch.meshgrid(off, off, indexing="ij"), -1).view(1, 2 * r + 1, 2 * r + 1, 2)
    outs = []
    for i, plane in enumerate(pyramid):
        hl, wl = plane.shape[-2:]
        c = pts / 2 ** i + delta
        gx = 2 * c[..., 0:1] / (wl - 1) - 1
        gy = 2 * c[..., 1:2] / (hl - 1) - 1
        s = F.grid_sample(plane, torch.cat([gx, gy], -1), align_corners=True)
        outs.append(s.view(b, h, w, -1))
    out = torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous()
    return out if out.dtype == torch.float64 else out.float()  # fp64 only for noise studies


def lookup_taps(coords: Tensor, level_sizes, radius=4):
    """Explicit fp32 replay of the sampler's index math (no grid_sample).

    Per level returns (x0, y0, wx, wy): int32 floor indices and fp32 fractional
    weights of shape (B*Q, 9) — x arrays are indexed by a (x-offset), y by b.
    Sequence: c = coord / 2^i + d ; g = 2*c/(n-1) - 1 ; u = ((g+1)/2)*(n-1) ;
    i0 = floor(u) ; w = u - i0   (utils.py:61-62 then ATen's align_corners
    unnormalize).  Every step is a separately rounded fp32 op.
    """
    b, _, h, w = coords.shape
    r = radius
    pts = coords.permute(0, 2, 3, 1).reshape(b * h * w, 2)
    off = torch.linspace(-r, r, 2 * r + 1)
    out = []
    for i, (hl, wl) in enumerate(level_sizes):
        res = []
        for axis, n in ((0, wl), (1, hl)):
            c = pts[:, axis:axis + 1] / 2 ** i + off[None, :]
            g = 2 * c / (n - 1) - 1
            u = ((g + 1) / 2) * (n - 1)
            i0 = torch.floor(u)
            res.append((i0.to(torch.int32), u - i0))
        out.append((res[0][0], res[1][0], res[0][1], res[1][1]))
    return out


def corr_lookup_explicit(pyramid: List[Tensor], coords: Tensor, radius=4) -> Tensor:
    """Gather-based lookup built on lookup_taps (zeros padding)."""
    b, _, h, w = coords.shape
    n = b * h * w
    k = 2 * radius + 1
    sizes = [tuple(p.shape[-2:]) for p in pyramid]
    outs = []
    for plane, (x0, y0, wx, wy) in zip(pyramid, lookup_taps(coords, sizes, radius)):
        hl, wl = plane.shape[-2:]
        flat = plane.reshape(n, hl * wl)
        acc = torch.zeros(n, k, k)
        for dy in (0, 1):
            for dx in (0, 1):
                xi = (x0 + dx).long()[:, :, None].expand(n, k, k)      # a-major
                yi = (y0 + dy).long()[:, None, :].expand(n, k, k)
                ok = (xi >= 0) & (xi < wl) & (yi >= 0) & (yi < hl)
                idx = (yi.clamp(0, hl - 1) * wl + xi.clamp(0, wl - 1)).reshape(n, k * k)
                v = torch.gather(flat, 1, idx).view(n, k, k) * ok
                wgt = (wx if dx else 1 - wx)[:, :, None] * (wy if dy else 1 - wy)[:, None, :]
                acc = acc + v * wgt
        outs.append(acc.view(b, h, w, k * k))
    return torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------
# update block + upsampling
# ----------------------------------------------------------------------------
def motion_encoder(sd, p, flow, corr):
    """update.py:89-97."""
    cor = torch.relu(_conv(sd, p + ".convc1", corr, 1, 0))
    cor = torch.relu(_conv(sd, p + ".convc2", cor, 1, 1))
    flo = torch.relu(_conv(sd, p + ".convf1", flow, 1, 3))
    flo = torch.relu(_conv(sd, p + ".convf2", flo, 1, 1))
    out = torch.relu(_conv(sd, p + ".conv", torch.cat([cor, flo], 1), 1, 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, p, h, x):
    """update.py:45-60: horizontal (1x5) pass then vertical (5x1) pass."""
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, p + ".convz" + tag, hx, 1, pad))
        r = torch.sigmoid(_conv(sd, p + ".convr" + tag, hx, 1, pad))
        q = torch.tanh(_conv(sd, p + ".convq" + tag, torch.cat([r * h, x], 1), 1, pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, p, net, inp, corr, flow):
    """update.py:126-135 → (net, up_mask, delta_flow)."""
    motion = motion_encoder(sd, p + ".encoder", flow, corr)
    net = sep_conv_gru(sd, p + ".gru", net, torch.cat([inp, motion], 1))
    delta = _conv(sd, p + ".flow_head.conv2", torch.relu(_conv(sd, p + ".flow_head.conv1", net, 1, 1)), 1, 1)
    up_mask = 0.25 * _conv(sd, p + ".mask.2", torch.relu(_conv(sd, p + ".mask.0", net, 1, 1)), 1, 0)
    return net, up_mask, delta


def upsample_flow(flow: Tensor, mask: Tensor) -> Tensor:
    """raft.py:159-170 convex 8x upsampling."""
    n, _, h, w = flow.shape
    wts = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    nb = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(wts * nb, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


# ----------------------------------------------------------------------------
# full model
# ----------------------------------------------------------------------------
def prepare_inputs(image1, image2, mask1, mask2, mask_channel=3):
    """ff_raft.py:31-38 ('point' masks) + :142-145 ([0,255] → [-1,1])."""
    assert mask1.shape[1] == 1
    if mask_channel != 1:
        mask1 = mask1.repeat(1, mask_channel, 1, 1)
    mask2 = torch.ones_like(mask1) * 255
    return tuple(2 * (t / 255.0) - 1.0 for t in (image1.contiguous(), image2.contiguous(), mask1, mask2))


def raft_forward(sd: Dict[str, Tensor], image1, image2, mask1, mask2, iters=12,
                 flow_init: Optional[Tensor] = None, test_mode=False, training=False,
                 fusion_type="1x1conv", prefix="", taps: Optional[dict] = None, corr_half: bool = False):
    """raft.py:173-236 on already-normalised inputs.

    ``taps`` (optional dict) receives intermediates for per-op parity tests.
    ``corr_half``: fp16 storage of the correlation pyramid (see corr_pyramid).
    """
    p = prefix
    fmap1 = cce_encoder(sd, p + "fnet", image1, mask1, "instance", training, fusion_type)
    fmap2 = cce_encoder(sd, p + "fnet", image2, mask2, "instance", training, fusion_type)
    if fmap1.dtype != torch.float64:  # raft.py:191-193; fp64 is kept for noise studies
        fmap1, fmap2 = fmap1.float(), fmap2.float()
    pyramid = corr_pyramid(corr_volume(fmap1, fmap2), half=corr_half)
    cnet = cce_encoder(sd, p + "cnet", image1, mask1, "batch", training, fusion_type)
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    b, _, hh, ww = image1.shape
    coords0 = coords_grid(b, hh // 8, ww // 8, image1.dtype)
    coords1 = coords0.clone()
    if flow_init is not None:
        coords1 = coords1 + flow_init
    if taps is not None:
        taps.update(fmap1=fmap1, fmap2=fmap2, cnet=cnet, pyramid=pyramid, iters=[])
    preds = []
    flow_up = None
    for _ in range(iters):
        coords1 = coords1.detach()
        corr = corr_lookup(pyramid, coords1)
        flow = coords1 - coords0
        net, up_mask, delta = update_block(sd, p + "update_block", net, inp, corr, flow)
        coords1 = coords1 + delta
        flow_up = upsample_flow(coords1 - coords0, up_mask)
        preds.append(flow_up)
        if taps is not None:
            taps["iters"].append(dict(corr=corr, net=net, up_mask=up_mask, delta=delta,
                                      coords1=coords1, flow_up=flow_up))
    if test_mode:
        return coords1 - coords0, flow_up
    return preds


def ffraft_forward(sd, image1, image2, mask1, mask2=None, raft_iters=12, flow_init=None,
                   test_mode=False, training=False, mask_channel=3, fusion_type="1x1conv",
                   taps=None, corr_half=False):
    """ff_raft.py:134-160 for use_fusion='parallel'; keys carry 'flow_net.'."""
    i1, i2, m1, m2 = prepare_inputs(image1, image2, mask1, mask2, mask_channel)
    return raft_forward(sd, i1, i2, m1, m2, raft_iters, flow_init, test_mode, training,
                        fusion_type, prefix="flow_net.", taps=taps, corr_half=corr_half)


# ----------------------------------------------------------------------------
# losses (losses/losses.py:18-130) — "next" row f1, used by the training parity test
# ----------------------------------------------------------------------------
def sequence_l1(preds, flow_gt, valid, gamma=0.8, max_flow=400.0):
    """EPELoss, losses.py:18-47."""
    n = len(preds)
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    ok = ((valid >= 0.5) & (mag < max_flow))[:, None].float()
    loss = 0.0
    for i, pr in enumerate(preds):
        loss = loss + gamma ** (n - i - 1) * (ok * (pr - flow_gt).abs()).mean()
    epe = torch.sum((preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[ok.view(-1) >= 0.5]
    return loss, {"epe": epe.mean().item()}


def synthetic_inputs(b, h, w, seed=0, n_points=500):
    """SURVEY §8d synthetic pair: randint images, ORB-like Bernoulli mask."""
    g = torch.Generator().manual_seed(seed)
    image1 = torch.randint(0, 256, (b, 3, h, w), generator=g).float()
    image2 = torch.randint(0, 256, (b, 3, h, w), generator=g).float()
    mask1 = (torch.rand(b, 1, h, w, generator=g) < n_points / (h * w)).float() * 255
    mask2 = torch.zeros_like(mask1)
    return image1, image2, mask1, mask2


def shifted_pair(b, h, w, seed=0, shift=(3, -5), n_points=500, smooth=4):
    """Structured pair: image2 = roll(image1, shift) + noise, images low-pass
    filtered so the correlation volume has real structure and lookups land at
    non-integer coordinates."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(b, 3, h // smooth + 2, w // smooth + 2, generator=g)
    image1 = F.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
    image2 = torch.roll(image1, shifts=shift, dims=(2, 3)) + torch.randn(b, 3, h, w, generator=g) * 2
    image2 = image2.clamp(0, 255)
    mask1 = (torch.rand(b, 1, h, w, generator=g) < n_points / (h * w)).float() * 255
    return image1.contiguous(), image2.contiguous(), mask1, torch.zeros_like(mask1)


def gaussian_box(kernel_size, sigma):
    """losses.py:7-15."""
    import numpy as np
    s3 = 3 * sigma
    xs = np.linspace(-s3, s3, kernel_size)
    x, y = np.meshgrid(xs, xs)
    gauss = 1 / (2 * np.pi * sigma ** 2) * np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return torch.FloatTensor((1 / gauss.sum()) * gauss).view(1, 1, kernel_size, kernel_size)


def sequence_loss(kind, preds, flow_gt, valid, mask, gamma=0.8, max_flow=400.0, kernel_size=5, sigma=1.7, lamda=0.8):
    """EPELoss / CPCL / MixLoss, losses.py:18-130 (kind in {'EPELoss','CPCL','MixLoss'})."""
    n = len(preds)
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    ok = (valid >= 0.5) & (mag < max_flow)
    if kind != "EPELoss":
        m = (mask > 0).float()
        pad = kernel_size // 2
        m = F.conv2d(F.pad(m, [pad, pad, pad, pad]), gaussian_box(kernel_size, sigma))
    loss = 0.0
    for i, pr in enumerate(preds):
        wgt = gamma ** (n - i - 1)
        l1 = (pr - flow_gt).abs()
        if kind == "CPCL":
            loss = loss + wgt * (ok[:, None] * m * l1).sum() / m.sum()
        elif kind == "MixLoss":
            loss = loss + lamda * wgt * (ok[:, None] * m * l1).sum() / m.sum()
            loss = loss + wgt * (ok[:, None] * l1).mean()
        else:
            loss = loss + wgt * (ok[:, None] * l1).mean()
    epe = torch.sum((preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[ok.view(-1)]
    return loss, {"epe": epe.mean().item(), "loss": float(loss.detach())}


def loss_inputs(seed=0, b=2, h=48, w=64, n=3):
    g = torch.Generator().manual_seed(seed)
    flow_gt = torch.randn(b, 2, h, w, generator=g) * 4
    flow_gt[0, :, :4, :4] = 500.0                       # beyond max_flow: excluded
    preds = [flow_gt + torch.randn(b, 2, h, w, generator=g) * (n - i) for i in range(n)]
    valid = (torch.rand(b, h, w, generator=g) > 0.1).float()
    mask = (torch.rand(b, 1, h, w, generator=g) < 0.05).float() * 255
    return preds, flow_gt, valid, mask


# ----------------------------------------------------------------------------
# init_mask modes (ff_raft.py:23-72)
# ----------------------------------------------------------------------------
def ellipse_element(k):
    """cv.getStructuringElement(MORPH_ELLIPSE, (k,k)) per OpenCV 4.7.0 (requirements.txt:
    opencv_python==4.7.0.72; cv2 is absent here -> published algorithm restated, PARITY UNPINNED)."""
    import numpy as np
    r = c = k // 2
    el = np.zeros((k, k), np.float32)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r and r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / float(r * r))))
            el[i, max(c - dx, 0):min(c + dx + 1, k)] = 1.0
    if k == 1:
        el[:] = 1.0
    return torch.from_numpy(el)


def init_mask(image1, image2, mask1, modal, mask_channel=3, dilate=31, kernel_size=31, kernel_sigma=5):
    """-> (mask1, mask2) in [0,255] as ff_raft.py:23-72 builds them."""
    if modal == "point":
        m1 = mask1.repeat(1, mask_channel, 1, 1)
        return m1, torch.ones_like(m1) * 255
    if modal == "frame":
        return image1.clone(), image2.clone()
    if modal == "neighborG":
        m = F.conv2d(mask1, gaussian_box(kernel_size, kernel_sigma), padding=kernel_size // 2)
        m = (m * 255 / m.max()).repeat(1, mask_channel, 1, 1)
        return m, torch.ones_like(m) * 255
    el = ellipse_element(dilate)[None, None]
    dil = F.conv2d(mask1 / 255, el, padding=dilate // 2) > 0
    if modal == "neighborE":
        m = (dil * 255).float().repeat(1, mask_channel, 1, 1)
        return m, torch.ones_like(m) * 255
    if modal == "context":
        return dil * image1, image2.clone()
    raise ValueError(modal)
