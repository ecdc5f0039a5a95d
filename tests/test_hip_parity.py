"""GPU parity: libfocusflow_hip (through its C ABI) against the CPU oracle and the
reference-generated golden vectors.  Everything here needs a real MI355X.

Tolerances: fp32 activations rtol 2e-5 (fp32 MFMA is an exact fma chain; the
summation order differs from oneDNN's); final flow max-abs 1e-3 (BASELINE.json
north_star); lookup tap indices bit-exact.
"""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_spec, load_golden
from oracle import ffraft_ref as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t, pad_to=None):
    """NCHW cpu -> NHWC device (test plumbing)."""
    t = t.permute(0, 2, 3, 1).contiguous()
    if pad_to and t.shape[-1] < pad_to:
        t = F.pad(t, (0, pad_to - t.shape[-1]))
    return t.to(DEV)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def close(a, b, rtol=2e-5, atol=None, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if atol is None:
        atol = rtol * max(1.0, float(np.abs(b).max()))
    err = np.abs(a - b) - rtol * np.abs(b)
    assert err.max() <= atol, f"{what}: max violation {err.max():.3e} (atol {atol:.3e}), max|ref| {np.abs(b).max():.3e}"


C5_FP16_VS_ORACLE = 2e-3   # px at 1/8 resolution: same definition on both sides; differences come from fp16 roundings that flip on 1e-6 volume noise
C5_FP16_VS_FP32 = 2e-2     # px on flow_low: effect of the fp16 storage itself on the 32-iteration flow (5.5e-3 measured on the oracle)


@pytest.fixture(scope="module")
def ops():
    from focusflow_official_amd import ops as _ops
    return _ops


def _cfg(ft="1x1conv"):
    return Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                     MODEL=Namespace(FUSION_TYPE=ft, LOAD_MODULE_TO_BRANCH=False))


def _model(sd, ft="1x1conv"):
    from focusflow_official_amd import FF_RAFT_FUSION
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg(ft))
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


# ----------------------------------------------------------------------------
# operators
# ----------------------------------------------------------------------------
CONV_CASES = [
    # (segments, cout, kh, kw, stride, pad, B, H, W, act, use_res)
    ([3], 64, 7, 7, 2, (3, 3), 2, 40, 56, 1, False),        # stem (Cin 3 -> padded 4)
    ([64], 64, 3, 3, 1, (1, 1), 2, 24, 40, 0, True),        # residual 3x3
    ([64], 96, 3, 3, 2, (1, 1), 1, 33, 47, 1, False),       # stride 2, odd sizes, Cout 96 tile
    ([64], 96, 1, 1, 2, (0, 0), 2, 32, 48, 0, False),       # downsample 1x1 s2
    ([128], 256, 1, 1, 1, (0, 0), 1, 16, 24, 0, True),      # fusion / output conv
    ([324], 256, 1, 1, 1, (0, 0), 1, 16, 24, 1, False),     # convc1 (K tail 324 = 10*32+4)
    ([2], 128, 7, 7, 1, (3, 3), 1, 16, 24, 1, False),       # convf1 (Cin 2 -> padded 4)
    ([192, 64], 126, 3, 3, 1, (1, 1), 1, 16, 24, 1, False), # motion conv: 2 segments, Cout 126
    ([128, 128, 128], 256, 1, 5, 1, (0, 2), 2, 16, 24, 2, False),  # convz|convr horizontal, sigmoid
    ([128, 128, 128], 128, 5, 1, 1, (2, 0), 1, 16, 24, 3, False),  # convq vertical, tanh
    ([256], 2, 3, 3, 1, (1, 1), 1, 16, 24, 0, False),       # flow head conv2 (Cout 2)
    ([256], 576, 1, 1, 1, (0, 0), 1, 16, 24, 0, False),     # mask head
    ([128], 128, 3, 3, 1, (1, 1), 8, 48, 64, 1, False),     # big-M: 128x128 tiles
    ([64], 64, 3, 3, 1, (1, 1), 2, 96, 128, 1, False),      # 128x64 tiles
]


@pytest.mark.parametrize("fmt", ["fp32", "f16x3", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"c{'+'.join(map(str, c[0]))}-o{c[1]}-k{c[2]}x{c[3]}-s{c[4]}")
def test_conv2d(ops, case, fmt):
    segs, cout, kh, kw, stride, pad, b, h, w, act, use_res = case
    g = torch.Generator().manual_seed(hash(str(case)) & 0xFFFF)
    cin = sum(segs)
    xs = [torch.randn(b, c, h, w, generator=g) for c in segs]
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(torch.cat(xs, 1), wt, bias, stride=stride, padding=pad)
    ref = [lambda v: v, torch.relu, torch.sigmoid, torch.tanh][act](ref)
    res = torch.randn_like(ref) if use_res else None
    if use_res:
        ref = torch.relu(ref + res)
    cin_pads = [(c + 3) // 4 * 4 for c in segs]
    assert len(segs) == 1 or cin_pads == segs
    wp = torch.empty(cout, kh * kw * sum(cin_pads), device=DEV)
    ops.pack_conv_weight(wt.to(DEV), wp, sum(cin_pads))
    # inputs as channel slices of a wider buffer to exercise ld != C
    xd = []
    for x, cp in zip(xs, cin_pads):
        buf = torch.zeros(b, h, w, cp + 8, device=DEV)
        buf[..., 4:4 + x.shape[1]] = nhwc(x)
        xd.append(buf[..., 4:4 + cp])
    w_fmt = {"fp32": 0, "f16x3": 1, "f16": 2}[fmt]
    if w_fmt:
        wp = ops.pack_split(wp)     # fp16-split rows for the half-precision matrix pipe
    out = ops.conv2d(xd, wp, bias.to(DEV), cout, kh, kw, stride, pad, act=act,
                     res=nhwc(res) if use_res else None, act_res=1 if use_res else 0, w_fmt=w_fmt)
    torch.cuda.synchronize()
    # f16x3 must hold the fp32 tolerance; plain fp16 operands are the reduced-precision mode
    close(nchw(out), ref, rtol=2e-5 if fmt != "f16" else 4e-3, what=f"conv2d[{fmt}]")


@pytest.mark.parametrize("b,h,w,cout", [(2, 40, 56, 64), (1, 41, 57, 64), (3, 96, 128, 64), (1, 384, 512, 64), (2, 30, 34, 48)])
def test_stem_conv_kernel(ops, b, h, w, cout):
    """conv_stem.hip: the encoders' Conv2d(3, 64, 7, stride 2, padding 3) (extractor.py:123) over NHWC4 input - persistent
    blocks, the patch split once, MFMA fragments read straight out of it - against F.conv2d at the fp32 tolerance of the
    f16x3 arithmetic: whole and ragged tiles, odd sizes, a channel count below the tile, eval-BatchNorm scale/shift +
    activation in the epilogue, and the InstanceNorm statistics of the output from the same launch."""
    g = torch.Generator().manual_seed(b * 100 + h)
    x = torch.randn(b, 3, h, w, generator=g)
    wt = torch.randn(cout, 3, 7, 7, generator=g) / 147 ** 0.5
    bias, sc, sh = (torch.randn(cout, generator=g) for _ in range(3))
    conv = F.conv2d(x, wt, bias, stride=2, padding=3)
    wp = torch.empty(cout, 49 * 4, device=DEV)
    ops.pack_conv_weight(wt.to(DEV), wp, 4)
    wp = ops.pack_split(wp)
    x4 = torch.zeros(b, h, w, 4, device=DEV)
    x4[..., :3] = nhwc(x)
    out, st = ops.conv2d([x4], wp, bias.to(DEV), cout, 7, 7, 2, (3, 3), w_fmt=1, want_stats=True)
    close(nchw(out), conv, what="stem conv")
    n = out.shape[1] * out.shape[2]
    ref = conv.double()
    close((st[..., 0] / n).cpu(), ref.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6, what="mean from the stem's epilogue")
    close((st[..., 1] / n - (st[..., 0] / n) ** 2).cpu(), ref.var(dim=(2, 3), unbiased=False), rtol=2e-4, atol=1e-7, what="variance from the stem's epilogue")
    out2 = ops.conv2d([x4], wp, bias.to(DEV), cout, 7, 7, 2, (3, 3), act=1, ch_scale=sc.to(DEV), ch_shift=sh.to(DEV), w_fmt=1)
    close(nchw(out2), torch.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), what="stem conv, scale/shift + relu")


def test_conv_epilogue_scale_shift_and_outscale(ops):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 16, 24, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24
    bias, sc, sh = (torch.randn(64, generator=g) for _ in range(3))
    ref = torch.relu((F.conv2d(x, wt, bias, padding=1) * 0.25) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    wp = torch.empty(64, 9 * 64, device=DEV)
    ops.pack_conv_weight(wt.to(DEV), wp, 64)
    out = ops.conv2d([nhwc(x)], wp, bias.to(DEV), 64, 3, 3, 1, 1, act=1, ch_scale=sc.to(DEV), ch_shift=sh.to(DEV),
                     out_scale=0.25)
    close(nchw(out), ref, what="epilogue")


@pytest.mark.parametrize("c,h,w,b", [(64, 64, 96, 2), (96, 31, 47, 3), (128, 16, 24, 1), (256, 16, 24, 2)])
def test_instance_norm(ops, c, h, w, b):
    g = torch.Generator().manual_seed(c)
    x = torch.randn(b, c, h, w, generator=g) * 3 + 1.5
    res = torch.randn(b, c, h, w, generator=g)
    xd = nhwc(x)
    st = ops.norm_stats(xd, per_sample=True)
    y = ops.norm_apply(xd, st, True, 1e-5, act=1)
    close(nchw(y), torch.relu(F.instance_norm(x)), rtol=1e-5, atol=2e-5, what="instance_norm+relu")
    y2 = ops.norm_apply(xd, st, True, 1e-5, act=1, res=nhwc(res))
    close(nchw(y2), torch.relu(res + torch.relu(F.instance_norm(x))), rtol=1e-5, atol=2e-5, what="instance_norm+res")


def test_batch_norm_train_and_eval(ops):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 96, 20, 28, generator=g) * 2 - 0.7
    bn = torch.nn.BatchNorm2d(96)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(96, generator=g))
        bn.bias.copy_(torch.randn(96, generator=g))
        bn.running_mean.copy_(torch.randn(96, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(96, generator=g) + 0.5)
    import copy
    bn_d = copy.deepcopy(bn).to(DEV)
    ref = bn(x)  # train mode: batch stats + running update
    xd = nhwc(x)
    st = ops.norm_stats(xd, per_sample=False)
    ops.bn_update_running(bn_d, st, 3 * 20 * 28)
    y = ops.norm_apply(xd, st, False, bn_d.eps, bn_d.weight, bn_d.bias)
    close(nchw(y), ref.detach(), rtol=1e-5, atol=3e-5, what="bn train")
    close(bn_d.running_mean.cpu(), bn.running_mean, rtol=1e-6, atol=1e-6, what="running_mean")
    close(bn_d.running_var.cpu(), bn.running_var, rtol=1e-5, atol=1e-6, what="running_var")
    bn.eval()
    sc, sh = ops.bn_fold(bn_d)
    close((xd * sc + sh).cpu().permute(0, 3, 1, 2), bn(x).detach(), rtol=1e-5, atol=3e-5, what="bn eval fold")


def test_prep_input_bit_exact(ops):
    i1, i2, m1, _ = orc.synthetic_inputs(2, 64, 96, seed=3)
    r1, r2, rm1, rm2 = orc.prepare_inputs(i1, i2, m1, None, 3)
    d = ops.prep_input(i1.to(DEV), 2, 64, 96, i1.to(DEV))
    assert torch.equal(nchw(d)[:, :3], r1) and (d[..., 3] == 0).all()
    dm = ops.prep_input(m1.to(DEV), 2, 64, 96, d)
    assert torch.equal(nchw(dm)[:, :3], rm1)
    dm2 = ops.prep_input(None, 2, 64, 96, d, fill=255.0)
    assert torch.equal(nchw(dm2)[:, :3], rm2)


# ----------------------------------------------------------------------------
# CorrBlock
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("b,h,w", [(2, 16, 24), (1, 46, 62), (1, 20, 16)])
def test_corr_volume_and_pyramid(ops, b, h, w):
    g = torch.Generator().manual_seed(h)
    f1, f2 = torch.randn(b, 256, h, w, generator=g), torch.randn(b, 256, h, w, generator=g)
    ref = orc.corr_pyramid(orc.corr_volume(f1, f2))
    vol = ops.corr_volume(nhwc(f1), nhwc(f2))
    pyr = ops.corr_pyramid(vol, h, w)
    for lv, (a, r) in enumerate(zip(pyr, ref)):
        assert a.shape[-2:] == r.shape[-2:]
        close(a.cpu(), r[:, 0], rtol=1e-5, atol=3e-5, what=f"pyramid level {lv}")


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("b,h,w", [(2, 16, 24), (1, 46, 62), (1, 20, 16), (1, 17, 19), (1, 68, 120)])
def test_corr_build_tiled_pyramid(ops, b, h, w, half):
    """ff_corr_build (one launch: f16x3 volume + the three pooled levels + tiled store) against the oracle's
    corr.py:12-27 restatement.  Level 0 to fp32 rounding (fp16 storage: to half an fp16 ulp more); the pooled levels
    must be EXACTLY ATen's avg_pool2d of the level below as stored (in fp16 mode: of the rounded level, rounded again)."""
    g = torch.Generator().manual_seed(h * 7 + w)
    f1, f2 = torch.randn(b, 256, h, w, generator=g), torch.randn(b, 256, h, w, generator=g)
    ref = orc.corr_pyramid(orc.corr_volume(f1, f2), half=half)
    pyr = ops.corr_build(nhwc(f1), nhwc(f2), half)
    assert pyr.half == half and pyr.levels[0].dtype == (torch.float16 if half else torch.float32)
    got = [pyr.rowmajor(l).cpu() for l in range(4)]
    for lv, (a, r) in enumerate(zip(got, ref)):
        assert a.shape[-2:] == r.shape[-2:]
        # fp16 storage: one rounding step of fp16 (2^-11 relative) on top of the fp32-level error of the volume
        close(a, r[:, 0], rtol=(1e-3 if half else 1e-5), atol=(1e-3 if half else 3e-5), what=f"tiled pyramid level {lv}")
    lvl = got[0][:, None]
    for lv in range(1, 4):
        lvl = torch.nn.functional.avg_pool2d(lvl, 2, stride=2)
        if half:
            lvl = lvl.half().float()
        assert torch.equal(lvl[:, 0], got[lv]), f"level {lv} is not the avg_pool2d of the stored level {lv - 1}"
    if half:
        assert torch.equal(got[0], got[0].half().float())


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_retile_round_trip_and_layout(ops, half):
    """ff_corr_retile: row-major -> tiled -> row-major is the identity (fp16: after one rounding), and the tiled
    offsets are the ones include/focusflow_hip.h documents."""
    h0, w0, n = 46, 62, 3
    g = torch.Generator().manual_seed(3)
    lv = [torch.randn(n, h0 >> l, w0 >> l, generator=g).to(DEV) for l in range(4)]
    pyr = ops.TiledPyramid.from_rowmajor(lv, half)
    th = 8 if half else 4
    for l in range(4):
        want = lv[l].half().float() if half else lv[l]
        assert torch.equal(pyr.rowmajor(l), want)
        h, w = h0 >> l, w0 >> l
        ntx = ((((w0 + 15) // 16 * 16) >> l) + 7) // 8
        nty = ((((h0 + 7) // 8 * 8) >> l) + th - 1) // th
        assert pyr.levels[l].shape[1] == ntx * nty * 8 * th
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        off = ((ys // th) * ntx + xs // 8) * (8 * th) + (ys % th) * 8 + xs % 8
        assert torch.equal(pyr.levels[l].cpu().float()[:, off.reshape(-1)], want.cpu().reshape(n, -1))


def _lookup_case(ops, pyr_cpu, coords_nchw, half=False):
    """pyr_cpu: list of (N,1,h,w) cpu planes; returns (hip_out NCHW, hip_taps, c_out, c_taps).  The product kernel
    (tiled pyramid) must agree BIT FOR BIT with the generic row-major kernel on the same values."""
    from oracle import corr_c
    lv = [p[:, 0].contiguous().to(DEV) for p in pyr_cpu]
    pyr = ops.TiledPyramid.from_rowmajor(lv, half)
    out, taps = ops.corr_lookup_tiled(pyr, nhwc(coords_nchw), want_taps=True)
    out_rm, taps_rm = ops.corr_lookup(lv, nhwc(coords_nchw), 4, want_taps=True)
    assert torch.equal(taps, taps_rm), "tiled and row-major lookups disagree on tap indices"
    assert torch.equal(out, out_rm), f"tiled vs row-major lookup: max diff {(out - out_rm).abs().max().item():.3e}"
    c_out, c_taps = corr_c.lookup([p[:, 0].numpy().copy() for p in pyr_cpu], coords_nchw.numpy())
    return nchw(out), taps.cpu().numpy(), c_out, c_taps


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_corr_block_in_batch_chunks_beyond_the_4gb_resource(half, monkeypatch):
    """The lookup kernel addresses a pyramid through one buffer resource (32-bit offsets): a batch whose pyramid would pass
    4 GB (configs[4] beyond 22 pairs) is built and looked up in batch chunks.  With the limit lowered so that 5 pairs make
    3 chunks, the lookups equal the unchunked block's bit for bit."""
    from focusflow_official_amd import corr_block
    g = torch.Generator().manual_seed(7)
    b, h, w = 5, 16, 24
    f1, f2 = (torch.randn(b, h, w, 256, generator=g).to(DEV) for _ in range(2))
    coords = (orc.coords_grid(b, h, w) + torch.rand(b, 2, h, w, generator=g) * 8 - 4).permute(0, 2, 3, 1).contiguous().to(DEV)
    with torch.no_grad():
        whole = corr_block.CorrBlock(f1, f2, pyramid_dtype="fp16" if half else "fp32")
        assert whole._chunks is None
        ref = whole(coords).clone()
        per_pair = h * w * sum(lv.shape[1] for lv in whole.pyr.levels) * (2 if half else 4)
        monkeypatch.setattr(corr_block, "_MAX_PYRAMID_BYTES", 2 * per_pair + 1)
        chunked = corr_block.CorrBlock(f1, f2, pyramid_dtype="fp16" if half else "fp32")
        assert chunked._chunks is not None and [(lo, hi) for lo, hi, _ in chunked._chunks] == [(0, 2), (2, 4), (4, 5)]
        assert torch.equal(chunked(coords), ref)
        for a, c in zip(chunked.corr_pyramid, whole.corr_pyramid):
            assert torch.equal(a, c)


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("h,w", [(48, 64), (46, 62), (16, 24), (17, 19), (68, 120)])
def test_lookup_taps_bit_exact_and_values(ops, h, w, half):
    g = torch.Generator().manual_seed(w)
    b = 1
    vol = torch.randn(b, h * w, h, w, generator=g) * 30
    pyr = orc.corr_pyramid(vol, half=half)    # fp16 storage: the lookup interpolates fp16-representable values in fp32
    base = orc.coords_grid(b, h, w)
    cases = {
        "integer": base.clone(),                                          # iteration 0: exactly on pixels
        "random": base + (torch.rand(base.shape, generator=g) * 16 - 8),
        "halves": base + 0.5, "eighths": base * 1.125,
        "far_outside": base + torch.tensor([w + 20.0, -h - 20.0]).view(1, 2, 1, 1),
        "edge": base + torch.tensor([-4.0, 4.0]).view(1, 2, 1, 1),
    }
    for name, c in cases.items():
        out, taps, c_out, c_taps = _lookup_case(ops, pyr, c, half)
        assert (taps == c_taps).all(), f"{name}: tap indices differ from the oracle in {(taps != c_taps).sum()} places"
        ref = orc.corr_lookup(pyr, c)  # the reference's grid_sample route
        close(out, ref, rtol=2e-6, atol=2e-4, what=f"lookup {name} vs grid_sample")
        close(out, c_out, rtol=1e-6, atol=1e-5, what=f"lookup {name} vs C oracle")
        # and the torch replay pinned to ATen's taps by tests/test_oracle_golden.py
        sizes = [tuple(p.shape[-2:]) for p in pyr]
        for lvl, (x0, y0, _, _) in enumerate(orc.lookup_taps(c, sizes)):
            assert (taps[:, lvl, 0] == x0.numpy()).all() and (taps[:, lvl, 1] == y0.numpy()).all()


def test_lookup_against_reference_vectors(ops, det_sd):
    """Reference outputs (golden) for a real pyramid: integer coords (iteration 0) and random coords."""
    g = load_golden("fwd_shift_128x192_b2_it12")
    inp = orc.shifted_pair(2, 128, 192, seed=1)
    taps = {}
    with torch.no_grad():
        orc.ffraft_forward(det_sd, *inp, raft_iters=1, test_mode=True, taps=taps)
    pyr_b0 = [p[:384] for p in taps["pyramid"]]
    out, _, _, _ = _lookup_case(ops, pyr_b0, torch.from_numpy(g["crand"]))
    close(out, g["look_rand"], rtol=2e-6, atol=1e-4, what="golden look_rand")
    out0, _, _, _ = _lookup_case(ops, pyr_b0, orc.coords_grid(1, 16, 24))
    close(out0, g["look0"], rtol=2e-6, atol=1e-4, what="golden look0")


# ----------------------------------------------------------------------------
# update-block glue
# ----------------------------------------------------------------------------
def test_gru_gates_and_coords(ops):
    g = torch.Generator().manual_seed(5)
    z, r, q, h = (torch.rand(2, 128, 16, 24, generator=g) for _ in range(4))
    zr = torch.cat([nhwc(z), nhwc(r)], -1)
    rh = ops.gru_rh(zr[..., 128:], nhwc(h))
    assert torch.equal(nchw(rh), r * h)
    hn = ops.gru_blend(zr[..., :128], nhwc(q), nhwc(h))
    close(nchw(hn), (1 - z) * h + z * q, rtol=0, atol=2e-7, what='gru blend')
    finit = torch.randn(2, 2, 16, 24, generator=g)
    c1 = ops.coords_init(2, 16, 24, zr, finit.to(DEV))
    ref_c1 = orc.coords_grid(2, 16, 24) + finit
    assert torch.equal(nchw(c1), ref_c1)
    delta = torch.randn(2, 2, 16, 24, generator=g)
    flow4 = torch.empty(2, 16, 24, 4, device=DEV)
    motion = torch.zeros(2, 16, 24, 128, device=DEV)
    ops.coords_step(c1, nhwc(delta), flow4, motion[..., 126:])
    ref_c1 = ref_c1 + delta
    assert torch.equal(nchw(c1), ref_c1)
    assert torch.equal(nchw(flow4)[:, :2], ref_c1 - orc.coords_grid(2, 16, 24)) and (flow4[..., 2:] == 0).all()
    assert torch.equal(motion[..., 126:], flow4[..., :2]) and (motion[..., :126] == 0).all()
    src = torch.randn(2, 256, 16, 24, generator=g)
    dst = torch.empty(2, 16, 24, 128, device=DEV)
    ops.act_copy(nhwc(src)[..., 128:], dst, 1)
    assert torch.equal(nchw(dst), torch.relu(src[:, 128:]))
    ops.act_copy(nhwc(src)[..., :128], dst, 3)
    close(nchw(dst), torch.tanh(src[:, :128]), rtol=1e-6, atol=1e-6, what="tanh")


def test_upsample_flow_known_answer(ops):
    g = load_golden("upsample")  # produced by the reference's RAFT.upsample_flow
    out = ops.upsample_flow(nhwc(torch.from_numpy(g["flow"])), nhwc(torch.from_numpy(g["mask"])))
    close(out.cpu(), g["out"], rtol=1e-6, atol=2e-6, what="upsample_flow")


# ----------------------------------------------------------------------------
# encoders and the full model
# ----------------------------------------------------------------------------
FWD = {
    "fwd_rand_128x192_b2_it12": (lambda: orc.synthetic_inputs(2, 128, 192, seed=0), 12),
    "fwd_shift_128x192_b2_it12": (lambda: orc.shifted_pair(2, 128, 192, seed=1), 12),
    "fwd_shift_128x160_b1_it4_init": (lambda: orc.shifted_pair(1, 128, 160, seed=2), 4),
}


@pytest.mark.parametrize("name", list(FWD))
def test_model_matches_reference_vectors(name, det_sd, ops):
    g = load_golden(name)
    make, iters = FWD[name]
    inp = [t.to(DEV) for t in make()]
    m = _model(det_sd)
    finit = torch.from_numpy(g["flow_init"]).to(DEV) if "flow_init" in g else None
    net = m.flow_net
    with torch.no_grad():
        b, _, h, w = inp[0].shape
        i1, i2 = ops.prep_input(inp[0], b, h, w, inp[0]), ops.prep_input(inp[1], b, h, w, inp[0])
        m1, m2 = ops.prep_input(inp[2], b, h, w, inp[0]), ops.prep_input(None, b, h, w, inp[0], fill=255.0)
        # deep fp32 stacks: error is relative to the tensor's scale (cnet has no output norm, |x| ~ 700)
        for what, got in (("fmap1", net.fnet(i1, m1)), ("fmap2", net.fnet(i2, m2)), ("cnet", net.cnet(i1, m1))):
            ref = g[what]
            close(nchw(got)[:, ::4], ref, rtol=2e-5, atol=1e-5 * float(np.abs(ref).max()), what=what)
        flow_low, flow_up = m(*inp, raft_iters=iters, flow_init=finit, test_mode=True)
        preds = m(*inp, raft_iters=iters, flow_init=finit)
    assert isinstance(preds, list) and len(preds) == int(g["n_preds"][0])
    assert flow_low.shape == g["flow_low"].shape and flow_up.shape == g["flow_up"].shape
    # BASELINE.json north_star: fp32 flow within 1e-3 max-abs of the reference
    close(flow_up.cpu(), g["flow_up"], rtol=0, atol=1e-3, what="flow_up")
    close(flow_low.cpu(), g["flow_low"], rtol=0, atol=1e-3, what="flow_low")
    close(preds[0].cpu(), g["pred_first"], rtol=0, atol=1e-3, what="pred[0]")
    close(preds[len(preds) // 2].cpu(), g["pred_mid"], rtol=0, atol=1e-3, what="pred[mid]")
    epe = np.sqrt(((flow_up.cpu().numpy() - g["flow_up"]) ** 2).sum(1)).mean()
    assert epe < 1e-3


def test_update_block_first_iteration(det_sd, ops):
    g = load_golden("fwd_shift_128x192_b2_it12")
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=1)]
    m = _model(det_sd)
    with torch.no_grad():
        preds = m(*inp, raft_iters=1)
    close(preds[0].cpu(), g["up1"], rtol=0, atol=2e-4, what="first-iteration flow_up")


def test_concat_fusion_variant(det_sd_concat):
    g = load_golden("fwd_concat_128x160_b1_it4")
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 128, 160, seed=8)]
    m = _model(det_sd_concat, "concat")
    with torch.no_grad():
        fl, fu = m(*inp, raft_iters=4, test_mode=True)
    close(fu.cpu(), g["flow_up"], rtol=0, atol=1e-3, what="concat flow_up")


@pytest.mark.parametrize("ft", ["SA", "CA"])
def test_attention_fusion_units_and_variants(ft, det_sd_sa, det_sd_ca):
    """SA / CA fusion (parallel_fusion.py:14-73): the unit alone against the reference's output on random (q, v),
    then the whole network against the reference's final flow."""
    from focusflow_official_amd import cce
    from oracle.weights import det_tensor
    g = load_golden(f"fwd_{ft.lower()}_128x160_b1_it4")
    unit = (cce._SA if ft == "SA" else cce._CA)(64)
    assert sorted(unit.state_dict()) == sorted(str(k) for k in g["unit_keys"])
    unit.load_state_dict({k: det_tensor(f"unit_{ft}." + k, v.shape) for k, v in unit.state_dict().items()})
    unit = unit.to(DEV)
    gen = torch.Generator().manual_seed(21)
    q, v = torch.randn(2, 64, 12, 20, generator=gen), torch.randn(2, 64, 12, 20, generator=gen)
    with torch.no_grad():
        out = unit.run(nhwc(q), nhwc(v))
    close(nchw(out), g["unit_out"], rtol=2e-5, what=f"{ft} unit")
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 128, 160, seed=8)]
    m = _model(det_sd_sa if ft == "SA" else det_sd_ca, ft)
    with torch.no_grad():
        fl, fu = m(*inp, raft_iters=4, test_mode=True)
    close(fu.cpu(), g["flow_up"], rtol=0, atol=1e-3, what=f"{ft} flow_up")


def test_config1_384x512_and_batch_consistency(det_sd):
    """BASELINE config 1 (B=1 384x512 it12) against the reference's vector, then
    config 2's size (B=8): eight copies of the pair must give eight identical flows
    equal to the B=1 flow (samples are independent — InstanceNorm, eval BatchNorm)."""
    g = load_golden("fwd_shift_384x512_b1_it12")
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 384, 512, seed=6)]
    m = _model(det_sd)
    with torch.no_grad():
        flow_low, flow_up = m(*inp, raft_iters=12, test_mode=True)
        close(flow_low.cpu(), g["flow_low"], rtol=0, atol=1e-3, what="384x512 flow_low")
        close(flow_up.cpu()[:, :, ::4, ::4], g["flow_up_sub"], rtol=0, atol=1e-3, what="384x512 flow_up")
        rep = [t.repeat(8, 1, 1, 1) for t in inp]
        fl8, fu8 = m(*rep, raft_iters=12, test_mode=True)
    assert fu8.shape == (8, 2, 384, 512)
    # not bit-identical: the dispatcher picks other tile shapes - and, for the single pair, K splits - for 3072 pixels than
    # for 24576 (different fp32 summation orders over 12 iterations); every copy inside the batch IS identical
    for i in range(8):
        close(fu8[i].cpu(), flow_up[0].cpu(), rtol=0, atol=5e-4, what=f"batch sample {i}")
        assert torch.equal(fu8[i], fu8[0])


def test_mask_stage_image_follows_an_in_place_weight_write(det_sd, monkeypatch):
    """`weight.data.mul_()` changes neither the parameter's version nor its storage; after
    cce.invalidate_packed the fused mask / up-sampling kernel must run on the NEW mask-head weights (its stage-major
    image is keyed on the PackedConv's pack generation) - compare it with the two-launch route, which repacks."""
    from focusflow_official_amd import cce, raft_net
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 128, 160, seed=12)]
    m = _model(det_sd)
    with torch.no_grad():
        before = m(*inp, raft_iters=3, test_mode=True)[1].clone()
        m.flow_net.update_block.mask[2].weight.data.mul_(1.5)
        m.flow_net.update_block.mask[2].bias.data.add_(0.1)
        assert cce.invalidate_packed(m) > 0
        fused = m(*inp, raft_iters=3, test_mode=True)[1].clone()
        monkeypatch.setattr(raft_net, "_MASK_UPSAMPLE", False)
        plain = m(*inp, raft_iters=3, test_mode=True)[1].clone()
    assert (fused - before).abs().max() > 1e-3, "the new mask-head weights must change the up-sampled flow"
    close(fused.cpu(), plain.cpu(), rtol=0, atol=2e-5, what="fused mask/up-sampling after invalidate_packed vs the two launches")


def test_config2_8_distinct_pairs_384x512_vs_oracle(det_sd):
    """The headline launch shapes on eight DIFFERENT pairs (8 x 48 x 64 = 24 576 queries per lookup, the launch the bench
    times): every sample of the batch against the CPU oracle run on that sample alone.  Eight copies of one pair
    (test_config1_384x512_and_batch_consistency) cannot see a cross-sample indexing slip - a sample reading its
    neighbour's pyramid plane, context features or coordinates gives the right answer there."""
    shifts = [(3, -5), (-4, 2), (1, 6), (-2, -3), (5, 1), (0, -7), (-6, 4), (2, 2)]
    pairs = [orc.shifted_pair(1, 384, 512, seed=40 + i, shift=sh) for i, sh in enumerate(shifts)]
    batch = [torch.cat([p[k] for p in pairs], 0).to(DEV) for k in range(4)]
    m = _model(det_sd)
    with torch.no_grad():
        fl8, fu8 = m(*batch, raft_iters=12, test_mode=True)
    fl8, fu8 = fl8.cpu(), fu8.cpu()
    means = []
    for i, p in enumerate(pairs):
        with torch.no_grad():
            ref_low, ref_up = orc.ffraft_forward(det_sd, *p, raft_iters=12, test_mode=True)
        close(fl8[i:i + 1], ref_low, rtol=0, atol=1e-3, what=f"distinct pair {i} flow_low")
        close(fu8[i:i + 1], ref_up, rtol=0, atol=1e-3, what=f"distinct pair {i} flow_up")
        means.append(ref_low)
    # the samples really are different problems: pairwise, their flows are further apart than the tolerance by far
    for i in range(8):
        for j in range(i):
            assert (means[i] - means[j]).abs().max() > 0.05, (i, j)


def test_reduced_precision_mode_end_to_end_epe_against_the_references_tf32_level(det_sd):
    """BASELINE configs[1] as literally written: reduced-precision arithmetic "vs reference EPE".  The throughput mode
    FF_CONV_PRECISION=f16 (one MFMA term: operands with 10 mantissa bits, fp32 accumulation) is the analogue of what the
    reference itself runs on a GPU - TF32, common.py:25-27 / ALLOW_TF32: true - whose cost in flow accuracy
    tests/golden/make_golden_tf32.py measured on the pinned oracle: 0.022 px mean / 0.21 px max end-point error against the
    fp32 reference on this input.  The HIP mode must stay within 3 x that, for one pair and for a batch of eight."""
    import json
    from focusflow_official_amd import ops as hops
    with open(os.path.join(os.path.dirname(__file__), "golden", "tf32_epe_384x512.json")) as f:
        tf32 = json.load(f)
    g = load_golden("fwd_shift_384x512_b1_it12")
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 384, 512, seed=6)]
    prev = hops.conv_precision()
    hops.set_conv_precision("f16")
    try:
        m = _model(det_sd)
        with torch.no_grad():
            fl1, fu1 = m(*inp, raft_iters=12, test_mode=True)
            fl8, fu8 = m(*[t.repeat(8, 1, 1, 1) for t in inp], raft_iters=12, test_mode=True)
    finally:
        hops.set_conv_precision(prev)
    ref_up, ref_low = torch.from_numpy(g["flow_up_sub"]), torch.from_numpy(g["flow_low"])

    def epe(a, b):
        return torch.sqrt(((a - b) ** 2).sum(1))
    for what, fu, fl in [("1 pair", fu1, fl1)] + [(f"sample {i} of 8", fu8[i:i + 1], fl8[i:i + 1]) for i in range(8)]:
        e_up, e_low = epe(fu.cpu()[:, :, ::4, ::4], ref_up), epe(fl.cpu(), ref_low)
        print(f"{what}: f16 mode EPE up mean {e_up.mean():.4f} max {e_up.max():.4f}, low mean {e_low.mean():.4f} max {e_low.max():.4f} px "
              f"(TF32 oracle: {tf32['epe_up_mean_px']:.4f} / {tf32['epe_up_max_px']:.4f})")
        assert torch.isfinite(fu).all()
        assert float(e_up.mean()) <= 3 * tf32["epe_up_mean_px"], what
        assert float(e_up.max()) <= 3 * tf32["epe_up_max_px"], what
        assert float(e_low.mean()) <= 3 * tf32["epe_low_mean_px"], what


def test_check_range_debug_mode_reports_activations_beyond_the_split_format(det_sd):
    """FF_CHECK_RANGE=1 / ops.CHECK_RANGE: the fp16-split conv formats read x as f16(4 x) + residual, so |x| >= 16376
    becomes inf - silently, apart from a NaN flow.  The debug mode measures max|x| of every forward conv input and the model
    raises at the end of the pass; the same weights pass cleanly, and with the check off nothing is measured.  The always-on
    guard (ops.guard_*) sees the same pass through its two probes: asynchronously while the model's history says the range
    is far away, synchronously (inside the forward) once it is not."""
    from focusflow_official_amd import _hip, ops as hops
    m = _model(det_sd)
    inp = [t.to(DEV) for t in orc.shifted_pair(1, 128, 160, seed=2)]
    hops.CHECK_RANGE = True
    try:
        with torch.no_grad():
            m(*inp, raft_iters=2, test_mode=True)                      # in range: no error (first forward on these weights: a careful one)
            m.check_range()
            assert m.flow_net._guard_hist and 1.0 < m.flow_net._guard_level < hops.X_LIMIT / hops.GUARD_MARGIN
            # an in-place change that the packed-weight caches follow by themselves (version counters): the guard keeps its history,
            # so the next forward is NOT careful
            m.flow_net.cnet.norm1.weight.mul_(2.0e4)                    # folded BatchNorm scale: activations of ~1e4 .. 1e5
            with pytest.raises(_hip.FocusFlowHipError, match="16376"):
                m(*inp, raft_iters=2, test_mode=True)                  # the debug mode raises at the end of the pass
    finally:
        hops.CHECK_RANGE = False
    # the always-on guard saw the same pass: it raises on demand or when the next forward starts - never silently returns the flow
    with pytest.raises(_hip.FocusFlowHipError, match="16376"):
        m.check_range()
    # ... and the model's running level is beyond the limit now: every forward looks at its words before it goes on
    with torch.no_grad(), pytest.raises(_hip.FocusFlowHipError, match="16376"):
        m(*inp, raft_iters=2, test_mode=True)
    hops._guard_state()["pending"].clear()
    monkey_guard = hops.RANGE_GUARD
    hops.RANGE_GUARD = False
    try:
        with torch.no_grad():
            m(*inp, raft_iters=2, test_mode=True)                      # guard off: nothing is measured, nothing raised
        m.check_range()
    finally:
        hops.RANGE_GUARD = monkey_guard


def test_context_output_beyond_the_split_range_is_repaired_on_the_exact_route(det_sd):
    """The reference has no activation range (fp32 everywhere, ff_raft.py:142-145 scales only the inputs).  A checkpoint whose
    context encoder ends in large values - here its last convolution scaled x 50: |x| ~ 35 000 against the split formats'
    16 376 - is caught by the always-on guard in the first forward on those weights, BEFORE the values meet a split
    convolution: the context features' share of the GRU gates (the only reader) runs on the exact-fp32 MFMA route, a warning is
    logged, the flow of that very forward matches the oracle; later forwards keep the route without further synchronisation."""
    from focusflow_official_amd import ops as hops
    sd = {k: v.clone() for k, v in det_sd.items()}
    for k in ("flow_net.cnet.conv2.weight", "flow_net.cnet.conv2.bias"):
        sd[k] *= 50.0
    m = _model(sd)
    inp = orc.shifted_pair(1, 128, 160, seed=3)
    with torch.no_grad():
        with pytest.warns(UserWarning, match="exact-fp32"):
            lo, up = m(*[t.to(DEV) for t in inp], raft_iters=4, test_mode=True)
        assert m.flow_net._exact_ctx and m.flow_net._guard_level >= hops.X_LIMIT
        ref_lo, ref_up = orc.ffraft_forward(sd, *inp, raft_iters=4, test_mode=True)
        close(up.cpu(), ref_up, rtol=0, atol=1e-3, what="flow_up with the repaired context share")
        close(lo.cpu(), ref_lo, rtol=0, atol=1e-3, what="flow_low with the repaired context share")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")                              # the second forward: same route, nothing new to say
            lo2, up2 = m(*[t.to(DEV) for t in inp], raft_iters=4, test_mode=True)
        assert torch.equal(up2, up)
    m.check_range()


def test_gru_steps_in_the_conv_epilogues_are_bit_identical(det_sd, monkeypatch):
    """Inference folds r * h into the epilogue of the z|r convolution and the state blend (1 - z) h + z tanh(q) into the q
    convolution's (FFConvParams.ep_mode; update.py:47-49).  The element-wise arithmetic is the same sequence of separately
    rounded operations as ff_gru_rh / ff_gru_blend, but the fused launches run the 16x16x32 variant of the patch kernel
    (another summation order inside the convolution than the 32x32x16 one the unfused path uses): flows equal to fp32
    rounding, and the fused op itself - same convolution kernel, epilogue on or off - bit for bit."""
    from focusflow_official_amd import update_block
    m = _model(det_sd)
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=4)]
    with torch.no_grad():
        monkeypatch.setattr(update_block, "_GRU_EPILOGUE", False)
        fl0, fu0 = m(*inp, raft_iters=6, test_mode=True)
        monkeypatch.setattr(update_block, "_GRU_EPILOGUE", True)
        fl1, fu1 = m(*inp, raft_iters=6, test_mode=True)
    close(fu1.cpu(), fu0.cpu(), rtol=0, atol=2e-4, what="flow_up, GRU epilogues on vs off")
    # the op against the element-wise kernels applied to the outputs of the un-fused convolutions
    from focusflow_official_amd import ops as hops
    gru = m.flow_net.update_block.gru
    g = torch.Generator().manual_seed(3)
    h = torch.tanh(torch.randn(2, 16, 24, 128, generator=g)).to(DEV)
    motion = torch.randn(2, 16, 24, 128, generator=g).to(DEV)
    ctx = torch.relu(torch.randn(2, 16, 24, 128, generator=g)).to(DEV)
    with torch.no_grad():
        (zr_pre, q_pre), _ = gru.prepare(ctx)
        zr_conv, q_conv = gru._zr_hm[0], gru._q_hm[0]
        zr_f = zr_conv([h, motion], res=zr_pre, act_res=hops.ACT_SIGMOID, ep_rh=h, ep_split=128)
        zr_u = zr_conv([h, motion], res=zr_pre, act_res=hops.ACT_SIGMOID)
        rh_u = hops.gru_rh(zr_u[..., 128:], h)
        h_f = q_conv([zr_f[..., 128:], motion], res=q_pre, act_res=hops.ACT_TANH, ep_blend=(zr_f[..., :128], h))
        q_u = q_conv([rh_u, motion], res=q_pre, act_res=hops.ACT_TANH)
        h_u = hops.gru_blend(zr_u[..., :128], q_u, h)
    close(zr_f[..., :128].cpu(), zr_u[..., :128].cpu(), rtol=0, atol=2e-6, what="z")
    close(zr_f[..., 128:].cpu(), rh_u.cpu(), rtol=0, atol=2e-6, what="r * h")
    close(h_f.cpu(), h_u.cpu(), rtol=0, atol=2e-6, what="new state")


def test_instance_norm_statistics_from_the_conv_epilogue(det_sd, monkeypatch):
    """Inference takes the InstanceNorm statistics of a convolution's output from that convolution's own epilogue
    (FFConvParams.stats_part -> ff_norm_stats_finish) instead of re-reading the output (extractor.py:48-56: every conv
    of the encoder feeds a norm).  The fused numbers - fp32 partial sums around a per-lane pivot, added up in double -
    must give the mean and variance of the double-precision pass over the same output, on whole and ragged tiles, on
    8-row and 4-row tile grids, with normalise-on-load inputs, and on a plane that is constant up to noise six
    orders of magnitude smaller (the case a one-pass sum of squares in fp32 gets wrong)."""
    from focusflow_official_amd import ops as hops, cce
    m = _model(det_sd)
    enc = m.flow_net.fnet
    blk = enc.layer1[0]
    g = torch.Generator().manual_seed(11)
    for (b, h, w) in [(2, 24, 40), (8, 64, 128), (1, 13, 21)]:
        x = torch.randn(b, h, w, 64, generator=g).to(DEV)
        with torch.no_grad():
            y, st = blk._p1(x, want_stats=True)
            ref = hops.norm_stats(y, per_sample=True)
            n = h * w
            mean, mean_r = st[..., 0] / n, ref[..., 0] / n
            var, var_r = st[..., 1] / n - mean ** 2, ref[..., 1] / n - mean_r ** 2
            close(mean.cpu(), mean_r.cpu(), rtol=1e-6, atol=1e-7, what=f"mean {b}x{h}x{w}")
            close(var.cpu(), var_r.cpu(), rtol=2e-6, atol=1e-9, what=f"variance {b}x{h}x{w}")
            # same statistics with the input normalised while loading
            sc, sh = hops.norm_coeffs(ref, n, 1e-5)
            y2, st2 = blk._p2(y, in_scale=sc, in_shift=sh, in_act=hops.ACT_RELU, want_stats=True)
            ref2 = hops.norm_stats(y2, per_sample=True)
            close((st2[..., 0] / n).cpu(), (ref2[..., 0] / n).cpu(), rtol=1e-6, atol=1e-7, what="mean, normalise-on-load")
            close((st2[..., 1] / n - (st2[..., 0] / n) ** 2).cpu(), (ref2[..., 1] / n - (ref2[..., 0] / n) ** 2).cpu(), rtol=2e-6, atol=1e-9,
                  what="variance, normalise-on-load")
    # nearly constant plane: a large bias over tiny variation
    conv = torch.nn.Conv2d(64, 64, 3, padding=1)
    with torch.no_grad():
        conv.weight.mul_(1e-4)
        conv.bias.fill_(100.0)
    pc = cce.PackedConv([conv.to(DEV)])
    x = torch.randn(2, 32, 48, 64, generator=g).to(DEV)
    with torch.no_grad():
        y, st = pc(x, want_stats=True)
        yd = y.double()
        var_true = yd.var(dim=(1, 2), unbiased=False)
        n = 32 * 48
        var = st[..., 1] / n - (st[..., 0] / n) ** 2
        assert float(var_true.max()) < 1e-4 and float(yd.mean()) > 99
        # E[x^2] - E[x]^2 in double from exact-enough partials: the cancellation costs ~1e4 * 2^-52 relative, nothing more
        close(var.cpu(), var_true.cpu(), rtol=0, atol=2e-11, what="variance of a nearly constant plane")   # 2^-52 * mean^2
    # end to end: on vs off
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=4)]
    with torch.no_grad():
        monkeypatch.setattr(hops, "CONV_STATS", False)
        _, fu0 = m(*inp, raft_iters=4, test_mode=True)
        monkeypatch.setattr(hops, "CONV_STATS", True)
        _, fu1 = m(*inp, raft_iters=4, test_mode=True)
    close(fu1.cpu(), fu0.cpu(), rtol=0, atol=2e-4, what="flow_up, statistics from the conv epilogue vs the statistics pass")


@pytest.mark.parametrize("b,h,w", [(2, 16, 24), (1, 13, 21), (8, 48, 64)])
def test_mask_conv_and_convex_upsampling_as_one_kernel(ops, b, h, w):
    """ff_mask_upsample_fwd: mask[2] (1x1 256 -> 576, x 0.25; update.py:121-124, :133), the soft-max over the nine
    neighbours and the convex combination (raft.py:159-170) in one launch, against the two launches it replaces
    (ff_conv2d_fwd + ff_upsample_flow - themselves pinned by reference vectors) and against torch: block-ragged pixel
    counts, image borders (zero-padded unfold), both split weight formats."""
    from focusflow_official_amd import cce
    g = torch.Generator().manual_seed(b * 10 + h)
    conv = torch.nn.Conv2d(256, 576, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(576, 256, 1, 1, generator=g) / 8)
        conv.bias.copy_(torch.randn(576, generator=g))
    hid = torch.relu(torch.randn(b, h, w, 512, generator=g)).to(DEV)
    flow = torch.randn(b, h, w, 4, generator=g).to(DEV) * 3
    flow[..., 2:] = 0
    pc = cce.PackedConv([conv.to(DEV)])
    with torch.no_grad():
        mask = pc(hid[..., 256:], out_scale=0.25)
        two = ops.upsample_flow(flow, mask)
        wq, bq = pc.get()
        one = ops.mask_upsample(hid[..., 256:], ops.mask_upsample_pack(wq), pc.fmt, bq, flow, 0.25)
        # torch: raft.py:159-170 on the fp32 convolution
        m = 0.25 * F.conv2d(nchw(hid[..., 256:].contiguous()).cpu(), conv.weight.cpu(), conv.bias.cpu())
        m = torch.softmax(m.view(b, 1, 9, 8, 8, h, w), dim=2)
        uf = F.unfold(8 * nchw(flow[..., :2].contiguous()).cpu(), [3, 3], padding=1).view(b, 2, 9, 1, 1, h, w)
        ref = torch.sum(m * uf, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(b, 2, 8 * h, 8 * w)
    close(one.cpu(), two.cpu(), rtol=0, atol=1e-4, what="one launch vs conv + upsample")
    close(one.cpu(), ref, rtol=0, atol=2e-4, what="one launch vs torch")
    # the one-term reduced-precision format through the same kernel (its TERMS = 1 instance): fp16 operand accuracy
    prev = ops.conv_precision()
    ops.set_conv_precision("f16")
    try:
        pc1 = cce.PackedConv([conv])
        with torch.no_grad():
            w1, b1 = pc1.get()
            assert pc1.fmt == 2
            one1 = ops.mask_upsample(hid[..., 256:], ops.mask_upsample_pack(w1), pc1.fmt, b1, flow, 0.25)
            two1 = ops.upsample_flow(flow, pc1(hid[..., 256:], out_scale=0.25))
    finally:
        ops.set_conv_precision(prev)
    close(one1.cpu(), two1.cpu(), rtol=0, atol=2e-3, what="one launch vs conv + upsample, f16 operands")
    close(one1.cpu(), ref, rtol=0, atol=0.25, what="one launch, f16 operands, vs torch")


def test_flow_head_takes_the_coordinate_step_bit_for_bit(ops):
    """FF_EP_COORDS: the 2-channel 3x3 flow head (update.py:13-14) with raft.py:223 / :219 in its epilogue
    (coords1 += delta, flow = coords1 - coords0) against the convolution followed by ff_coords_step: same delta, same
    coordinates, same flow4, bit for bit."""
    from focusflow_official_amd import cce
    g = torch.Generator().manual_seed(21)
    conv = torch.nn.Conv2d(256, 2, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(2, 256, 3, 3, generator=g) / 48)
    pc = cce.PackedConv([conv.to(DEV)])
    b, h, w = 2, 17, 29
    x = torch.relu(torch.randn(b, h, w, 256, generator=g)).to(DEV)
    c0 = ops.coords_init(b, h, w, x)
    c0 += (torch.randn(c0.shape, generator=g) * 3).to(DEV)       # test plumbing only
    ca, cb = c0.clone(), c0.clone()
    with torch.no_grad():
        d1 = pc(x)
        f1 = ops.empty_nhwc(b, h, w, 4, x)
        ops.coords_step(ca, d1, f1, None)
        f2 = ops.empty_nhwc(b, h, w, 4, x)
        d2 = pc(x, ep_coords=(cb, f2))
    assert pc.fmt == 0      # fp32 rows: the vector-ALU kernel
    assert torch.equal(d1, d2) and torch.equal(ca, cb) and torch.equal(f1, f2)


def test_launch_timing_counts_the_lookup_dispatches(ops):
    """ff_launch_timing_begin / _end: HIP events bound to each lookup dispatch (hipExtLaunchKernelGGL) - the count is the
    number of launches in between, the durations are those of a real kernel (a few to a few hundred microseconds), and
    nothing is timed once it is switched off."""
    g = torch.Generator().manual_seed(2)
    b, h, w = 2, 32, 40
    f12 = torch.randn(2 * b, h, w, 256, generator=g).to(DEV)
    pyr = ops.corr_build(f12[:b].contiguous(), f12[b:].contiguous())
    coords = ops.coords_init(b, h, w, f12)
    ops.launch_timing_begin(ops.TIME_LOOKUP)
    for _ in range(5):
        out = ops.corr_lookup_tiled(pyr, coords)
    n, total, lo, hi = ops.launch_timing_end(ops.TIME_LOOKUP)
    assert n == 5 and 0.5 < lo <= hi < 2000 and abs(total - 5 * (total / 5)) < 1e-6 and lo * 5 <= total * 1.0001 <= hi * 5 * 1.0001
    ops.corr_lookup_tiled(pyr, coords)
    assert ops.launch_timing_end(ops.TIME_LOOKUP)[0] == 0
    assert torch.isfinite(out).all()


def test_skip_unused_upsample_is_bit_identical(det_sd):
    """Opt-in inference shortcut: mask head + convex up-sampling for the last iteration only (the reference throws
    the other results away in test_mode, raft.py:226-236) must not change a single bit of either output."""
    m = _model(det_sd)
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=1)]
    with torch.no_grad():
        fl0, fu0 = m(*inp, raft_iters=5, test_mode=True)
        m.flow_net.skip_unused_upsample = True
        fl1, fu1 = m(*inp, raft_iters=5, test_mode=True)
        m.flow_net.skip_unused_upsample = False
    assert torch.equal(fl0, fl1) and torch.equal(fu0, fu1)


def test_gru_context_share_computed_once_matches_per_iteration_convs(det_sd, monkeypatch):
    """Inference computes the context features' share of the six GRU gate convolutions once per forward
    (SepConvGRU.prepare: x = cat[inp, motion] and inp never changes, update.py:132) and adds it in the epilogue of
    the per-iteration convolutions over [h, motion].  Same arithmetic up to the summation order: the flows must agree
    with the full 384-channel convolutions to fp32 rounding."""
    from focusflow_official_amd import raft_net
    m = _model(det_sd)
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=4)]
    with torch.no_grad():
        monkeypatch.setattr(raft_net, "_GRU_CTX_ONCE", False)
        fl0, fu0 = m(*inp, raft_iters=6, test_mode=True)
        monkeypatch.setattr(raft_net, "_GRU_CTX_ONCE", True)
        fl1, fu1 = m(*inp, raft_iters=6, test_mode=True)
    close(fl1.cpu(), fl0.cpu(), rtol=0, atol=2e-5, what="flow_low")
    close(fu1.cpu(), fu0.cpu(), rtol=0, atol=1e-4, what="flow_up")


@pytest.mark.parametrize("b,c,h,w", [(2, 64, 40, 48), (16, 64, 64, 128), (3, 96, 31, 47), (2, 128, 16, 24)])
def test_conv_normalises_its_input_while_loading(ops, b, c, h, w):
    """FFConvParams.in_scale / in_shift: conv(relu(instance_norm(x))) with the normalisation applied inside the conv's
    loader (ff_norm_coeffs tables) must give the bits of norm_apply followed by the plain conv - same coefficients,
    same fp32 operations, zero padding after the normalisation."""
    g = torch.Generator().manual_seed(b * 100 + c)
    x = nhwc(torch.randn(b, c, h, w, generator=g) * 2 + 0.7)
    wt = torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5
    bias = torch.randn(c, generator=g)
    wp = torch.empty(c, 9 * c, device=DEV)
    ops.pack_conv_weight(wt.to(DEV), wp, c)
    wp = ops.pack_split(wp)
    st = ops.norm_stats(x, per_sample=True)
    ref = ops.conv2d([ops.norm_apply(x, st, True, 1e-5, act=1)], wp, bias.to(DEV), c, 3, 3, 1, 1, w_fmt=1)
    sc, sh = ops.norm_coeffs(st, h * w, 1e-5)
    out = ops.conv2d([x], wp, bias.to(DEV), c, 3, 3, 1, 1, w_fmt=1, in_scale=sc, in_shift=sh, in_act=1)
    assert torch.equal(out, ref)


def test_normalise_on_load_forward_is_bit_identical(det_sd, monkeypatch):
    from focusflow_official_amd import cce
    m = _model(det_sd)
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=6)]
    with torch.no_grad():
        monkeypatch.setattr(cce, "_NORM_ON_LOAD", False)
        fl0, fu0 = m(*inp, raft_iters=3, test_mode=True)
        monkeypatch.setattr(cce, "_NORM_ON_LOAD", True)
        fl1, fu1 = m(*inp, raft_iters=3, test_mode=True)
    assert torch.equal(fl0, fl1) and torch.equal(fu0, fu1)


def test_paired_fusion_convs_are_bit_identical(det_sd, monkeypatch):
    """Inference runs both 1x1 convs of a FusionUnit (parallel_fusion.py:142-150) as one launch over the segments
    [img, mask] with an anti-diagonal weight and two residuals (FFConvParams.res2).  The added products are exact zeros
    in whole K chunks, so the result must be bit-identical to the two separate launches."""
    from focusflow_official_amd import cce
    inp = [t.to(DEV) for t in orc.shifted_pair(2, 128, 192, seed=1)]
    m = _model(det_sd)
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(cce, "_PAIR_FUSION", flag)
        with torch.no_grad():
            outs.append(m(*inp, raft_iters=2, test_mode=True))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_config5_shape_540x960_padded(det_sd):
    """BASELINE config 5's frame size: 540x960 replicate-padded to 544x960 (68x120 at 1/8: level 3 is 8x15, odd
    widths at two pyramid levels), against the CPU oracle.  Few iterations on purpose: with synthetic weights the
    recurrence is not contractive at this size — the reference's own arithmetic run in fp32 and in fp64 differs by
    3e-3 px after 12 iterations and 0.45 px after 32 (tests/diagnostics/check_c5.py) — so parity is asserted where
    rounding noise has not been amplified yet."""
    from focusflow_official_amd.utils import InputPadder
    inp = orc.shifted_pair(1, 540, 960, seed=3)
    pad = InputPadder(inp[0].shape)
    pin = pad.pad(*inp)
    assert pin[0].shape[-2:] == (544, 960)
    m = _model(det_sd)
    with torch.no_grad():
        fl, fu = m(*[t.to(DEV) for t in pin], raft_iters=3, test_mode=True)
        rl, ru = orc.ffraft_forward(det_sd, *pin, raft_iters=3, test_mode=True)
    close(fl.cpu(), rl, rtol=0, atol=1e-3, what="C5 flow_low")
    close(pad.unpad(fu.cpu()), pad.unpad(ru), rtol=0, atol=1e-3, what="C5 flow_up")


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_full_size_lookup_properties(ops, half):
    """Size-independent properties at B=8, 48x64 (BASELINE config 2 shapes), product path (ff_corr_build + tiled lookup)."""
    g = torch.Generator().manual_seed(0)
    b, h, w = 8, 48, 64
    f1 = torch.randn(b, h, w, 256, generator=g).to(DEV)
    f2 = torch.randn(b, h, w, 256, generator=g).to(DEV)
    pyr = ops.corr_build(f1, f2, half)
    lv = [pyr.rowmajor(l) for l in range(4)]
    vol = lv[0].view(b, h * w, h * w)
    coords = ops.coords_init(b, h, w, f1)
    out = ops.corr_lookup_tiled(pyr, coords)
    # the same volume through the grouped-conv route (other tile shapes, other summation order)
    close(vol.cpu(), ops.corr_volume(f1, f2).cpu(), rtol=(1e-3 if half else 1e-5), atol=(1e-3 if half else 3e-5), what="volume, two routes")
    # centre tap (k=40) of level 0 at integer coords is the volume's own diagonal entry
    diag = vol.diagonal(dim1=1, dim2=2)
    # (not bit-equal: at W=64 the sampler's fp32 normalise/un-normalise round trip moves some
    #  integer coordinates by an ulp, exactly as in the reference — SURVEY §7 "hard parts")
    close(out.view(b, h * w, 324)[..., 40].cpu(), diag.cpu(), rtol=0, atol=(2e-2 if half else 1e-3), what="centre tap")
    # linearity of the volume in fmap1 (not bit-exact: the fp16 residual of a split operand is a subnormal half for
    # small values, so doubling the operand changes its last bits)
    vol2 = ops.corr_build(f1 * 2, f2, half).rowmajor(0).view(b, h * w, h * w)
    close(vol2.cpu(), (vol * 2).cpu(), rtol=(1e-3 if half else 1e-6), atol=(2e-3 if half else 1e-5), what="linearity")
    # pyramid means are preserved level to level (48x64 divides evenly)
    for lo, hi in zip(lv[:-1], lv[1:]):
        close(hi.mean(dim=(1, 2)).cpu(), lo.mean(dim=(1, 2)).cpu(), rtol=1e-4, atol=(2e-3 if half else 1e-4), what="pool mean")
    # the tiled lookup agrees bit for bit with the row-major kernel on the stored values
    assert torch.equal(out, ops.corr_lookup(lv, coords, 4))


def _c5_state_dict():
    from oracle.weights import det_tensor
    return {k: det_tensor(k, s, flow_head_damp=0.01) for k, s, _ in golden_spec()}


def test_baseline_config5_544x960_it32_fp32_pyramid():
    """BASELINE configs[4] at full size and full length: 540x960 padded to 544x960, 32 iterations, against the
    REFERENCE's own fp32 run (tests/golden/make_golden_c5.py: raft.py:173-236; flow_head.conv2 damped so the recurrence
    is contractive - the spread between that run and the fp64 evaluation of the same arithmetic is 4e-4 px).  Bounds:
    1e-3 px against the fp32 reference (north-star tolerance), and the HIP path must not sit further from the fp64
    evaluation than twice the reference's own fp32 run does (+1e-4)."""
    g = load_golden("fwd_c5_544x960_b1_it32")
    inp = orc.shifted_pair(1, 544, 960, seed=3)
    m = _model(_c5_state_dict())
    with torch.no_grad():
        fl, fu = m(*[t.to(DEV) for t in inp], raft_iters=32, test_mode=True)
    fl, fu = fl.cpu().numpy(), fu.cpu().numpy()[:, :, ::4, ::4]
    assert np.abs(g["flow_low_fp64"]).max() > 1.0
    close(fl, g["flow_low_fp32"], rtol=0, atol=1e-3, what="C5 it32 flow_low vs reference fp32")
    close(fu, g["flow_up_sub_fp32"], rtol=0, atol=1e-3, what="C5 it32 flow_up vs reference fp32")
    ref_spread = np.abs(g["flow_up_sub_fp32"].astype(np.float64) - g["flow_up_sub_fp64"]).max()
    hip_spread = np.abs(fu.astype(np.float64) - g["flow_up_sub_fp64"]).max()
    assert hip_spread <= 2 * ref_spread + 1e-4, (hip_spread, ref_spread)


def test_baseline_config5_544x960_it32_fp16_pyramid():
    """The same run with the correlation pyramid STORED in fp16 (BASELINE configs[4]).  The reference has no such mode
    on its CPU path; the oracle restates what torch autocast gives CorrBlock (oracle.corr_pyramid(half=True)), the
    per-op tests pin the kernels to that definition bit for bit, and here the whole 32-iteration forward must (a) match
    the oracle run in that mode and (b) stay near the reference's fp32 fixture: fp16 rounds every correlation value to
    2^-11 relative, which moves the flow by a few 1e-3 px on this input (bounds measured on the oracle: see DESIGN.md)."""
    g = load_golden("fwd_c5_544x960_b1_it32")
    inp = orc.shifted_pair(1, 544, 960, seed=3)
    sd = _c5_state_dict()
    m = _model(sd)
    m.flow_net.corr_pyramid_dtype = "fp16"
    with torch.no_grad():
        fl, fu = m(*[t.to(DEV) for t in inp], raft_iters=32, test_mode=True)
        rl, ru = orc.ffraft_forward(sd, *inp, raft_iters=32, test_mode=True, corr_half=True)
    close(fl.cpu(), rl, rtol=0, atol=C5_FP16_VS_ORACLE, what="C5 it32 fp16 pyramid: flow_low vs oracle (fp16 pyramid)")
    close(fu.cpu(), ru, rtol=0, atol=8 * C5_FP16_VS_ORACLE, what="C5 it32 fp16 pyramid: flow_up vs oracle (fp16 pyramid)")   # flow_up = 8 x flow
    close(fl.cpu(), g["flow_low_fp32"], rtol=0, atol=C5_FP16_VS_FP32, what="C5 it32 fp16 pyramid vs reference fp32")


def test_hipgraph_replay_matches_eager(det_sd):
    """The captured forward (focusflow_official_amd/graph.py) replays bit-identically and tracks new inputs.  Against the
    eager forward it agrees to fp32 summation order only: while a graph is captured - no host time per launch at stake -
    the 1/8-resolution convolutions of a single pair take K splits that the eager path leaves alone (ops.conv2d)."""
    from focusflow_official_amd.graph import GraphedForward
    m = _model(det_sd)
    a = [t.to(DEV) for t in orc.shifted_pair(1, 128, 192, seed=31)]
    b = [t.to(DEV) for t in orc.shifted_pair(1, 128, 192, seed=32)]
    with torch.no_grad():
        ea = [t.clone() for t in m(*a, raft_iters=4, test_mode=True)]
        eb = [t.clone() for t in m(*b, raft_iters=4, test_mode=True)]
    gf = GraphedForward(m, a, raft_iters=4)
    ga = [t.clone() for t in gf(*a)]
    gb = [t.clone() for t in gf(*b)]
    ga2 = [t.clone() for t in gf(*a)]
    torch.cuda.synchronize()
    for x, y in zip(ea, ga):
        close(y.cpu(), x.cpu(), rtol=0, atol=2e-4, what="graph vs eager (input a)")
    for x, y in zip(eb, gb):
        close(y.cpu(), x.cpu(), rtol=0, atol=2e-4, what="graph vs eager (input b)")
    for x, y in zip(ga, ga2):
        assert torch.equal(x, y)
    assert not torch.equal(ga[1], gb[1])


@pytest.mark.parametrize("modal", ["frame", "neighborG", "neighborE", "context"])
def test_mask_modes(modal, det_sd):
    """init_mask modes (ff_raft.py:23-72): prepared mask tensors and the resulting flow vs the oracle."""
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd import ops as _ops
    cfg = _cfg()
    cfg.TRAIN.MASK_MODAL, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.KERNEL_SIZE, cfg.TRAIN.KERNEL_SIGMA = modal, 31, 31, 5
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    m.load_state_dict(det_sd)
    m = m.to(DEV).eval()
    inp = orc.shifted_pair(2, 128, 160, seed=17)
    r1, r2 = orc.init_mask(inp[0], inp[1], inp[2], modal)
    ref_in = tuple(2 * (t / 255.0) - 1.0 for t in (inp[0], inp[1], r1, r2))
    if modal in MASK_TABLE_MODES:
        from focusflow_official_amd.model import MASK_MODES, ellipse_table, gaussian_table
        tab = gaussian_table(31, 5) if modal == "neighborG" else ellipse_table(31)
        got = _ops.mask_prepare(MASK_MODES[modal], inp[2].to(DEV), inp[0].to(DEV), tab.to(DEV))
        close(nchw(got)[:, :3], ref_in[2], rtol=0, atol=2e-5, what=f"{modal} mask tensor")
    with torch.no_grad():
        _, fu = m(*[t.to(DEV) for t in inp], raft_iters=3, test_mode=True)
        _, ref = orc.raft_forward(det_sd, *ref_in, iters=3, test_mode=True, prefix="flow_net.")
    close(fu.cpu(), ref, rtol=0, atol=1e-3, what=f"{modal} flow")


@pytest.mark.parametrize("modal", ["point", "frame", "neighborG"])
def test_wrapper_mask_modes_match_reference_vectors(modal, det_sd):
    """The whole drop-in call FF_RAFT_FUSION(image1, image2, mask1, mask2) - init_mask, input scaling, RAFT - against
    vectors from the reference's own wrapper class (ff_raft.py:134-164) for the modes it can run here."""
    from focusflow_official_amd import FF_RAFT_FUSION
    g = load_golden("wrapper_modes_128x160_b2_it3")
    cfg = _cfg()
    cfg.TRAIN.MASK_MODAL, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.KERNEL_SIZE, cfg.TRAIN.KERNEL_SIGMA = modal, 31, 31, 5
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    m.load_state_dict(det_sd)
    m = m.to(DEV).eval()
    inp = orc.shifted_pair(2, 128, 160, seed=17)
    with torch.no_grad():
        fl, fu = m(*[t.to(DEV) for t in inp], raft_iters=3, test_mode=True)
    close(fl.cpu(), g[f"{modal}_flow_low"], rtol=0, atol=1e-3, what=f"{modal} flow_low")
    close(fu.cpu(), g[f"{modal}_flow_up"], rtol=0, atol=1e-3, what=f"{modal} flow_up")


MASK_TABLE_MODES = ("neighborG", "neighborE", "context")


def test_input_padder_roundtrip():
    from focusflow_official_amd.utils import InputPadder
    x = torch.arange(2 * 3 * 436 * 1022, dtype=torch.float32).view(2, 3, 436, 1022)
    for mode in ("sintel", "kitti"):
        p = InputPadder(x.shape, mode)
        (y,) = p.pad(x)
        assert y.shape[-2] % 8 == 0 and y.shape[-1] % 8 == 0 and torch.equal(p.unpad(y), x)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,h,w,k,res", [(672, 32, 14, 32, (3, 3), False), (416, 96, 7, 16, (3, 3), True), (544, 128, 12, 20, (1, 5), False)])
def test_split_k_convolution_small_plane_long_reduction(cin, cout, h, w, k, res):
    """FFConvParams.splitk (ff_conv2d_splitk_hint): partial sums of K ranges by separate blocks, added in a fixed order
    by the finishing launch - same result as torch's conv to fp32 rounding, identical from run to run."""
    import torch.nn.functional as F
    from focusflow_official_amd import ops, _hip
    g = torch.Generator().manual_seed(cin + cout)
    kh, kw = k
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    bias = torch.randn(cout, generator=g)
    r = torch.randn(1, cout, h, w, generator=g) if res else None
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=(kh // 2, kw // 2)), 0.1)
    if res:
        ref = torch.relu(ref + r.double())
    rows = torch.empty((cout, kh * kw * cin), device="cuda")
    ops.pack_conv_weight(wt.cuda(), rows, cin, 0)
    wsplit = ops.pack_split(rows)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rd = r.permute(0, 2, 3, 1).contiguous().cuda() if res else None
    # the library must want to split this shape (otherwise the test tests nothing)
    p = _hip.FFConvParams()
    p.x[0], p.x_ld[0], p.x_c[0] = xd.data_ptr(), cin, cin
    p.groups, p.B, p.H, p.W, p.Ho, p.Wo, p.Cout = 1, 1, h, w, h, w, cout
    p.KH, p.KW, p.stride, p.pad_h, p.pad_w, p.w_format = kh, kw, 1, kh // 2, kw // 2, _hip.W_F16X3
    assert _hip.load().ff_conv2d_splitk_hint(p) >= 2
    outs = [ops.conv2d([xd], wsplit, bias.cuda(), cout, kh, kw, 1, (kh // 2, kw // 2), act=ops.ACT_LEAKY, res=rd,
                       act_res=ops.ACT_RELU if res else ops.ACT_NONE, w_fmt=_hip.W_F16X3) for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    got = outs[0].permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_memory_probe_moves_the_bytes_it_reports():
    """ff_probe_memory_kernel (bench.py's memory-only companion of the lookup): every stored byte comes from the source
    buffer, the byte counts are those of the launch shape, and its launches are timed under FF_TIME_PROBE."""
    from focusflow_official_amd import ops
    blocks, trips = 64, 3
    src = torch.full((1 << 20,), 7, dtype=torch.uint8, device=DEV)
    dst = torch.zeros(blocks * trips * 4096 + 16, dtype=torch.uint8, device=DEV)
    ops.launch_timing_begin(ops.TIME_PROBE)
    rd, wr = ops.probe_memory_kernel(src, dst, 128, blocks, trips, 5)
    n, tot, lo, hi = ops.launch_timing_end(ops.TIME_PROBE)
    assert (rd, wr) == (blocks * trips * 5120, blocks * trips * 4096) and n == 1 and 0 < lo <= hi
    assert bool((dst[:wr] == 7).all()) and bool((dst[wr:] == 0).all())
    with pytest.raises(Exception, match="dst holds"):
        ops.probe_memory_kernel(src, dst[:1000], 128, blocks, trips, 5)
