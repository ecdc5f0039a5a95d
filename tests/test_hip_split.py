"""GPU tests of the split-pair activation format (FF_FMT_SPLIT) and of the LDS-DMA convolution that reads it
(csrc/conv_dma.hip): the update block's inference path since round 4.

The format is defined so that a convolution over a split-pair tensor computes the same bits as the convolution over the
fp32 tensor (the producer writes what the consumer's loader would have made of the fp32 value).  Held here to:
  * the bytes of the format itself (numpy emulation of x0 = fp16(4 v), x1 = fp16(4 v - x0));
  * torch.equal against the fp32-input kernel where both run the same matrix instruction (the GRU-epilogue instances of
    conv_patch.hip), 2e-6 relative where the summation order inside a 32-channel chunk differs, 2e-5 against F.conv2d;
  * every tile shape the dispatcher can choose, ragged planes, 1-3 input segments, channel counts off the tile;
  * the whole forward: split-pair activations on / off give the same flow (1e-4 px; 5e-4 px after 12 iterations at 384 x 512), both within 1e-3 px of the oracle.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ffraft_ref as orc
from test_hip_parity import DEV, close, nchw, nhwc, _model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from focusflow_official_amd import ops as _ops
    return _ops


def _split_np(v):
    """numpy emulation of the format: (x0, x1) as float16 arrays."""
    sv = (v.astype(np.float32) * np.float32(4.0)).astype(np.float32)
    x0 = sv.astype(np.float16)
    x1 = (sv - x0.astype(np.float32)).astype(np.float16)
    return x0, x1


def test_split_copy_bytes_and_round_trip(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 7, 96, generator=g) * 3
    x[0, 0, 0, :8] = torch.tensor([0.0, 0.5, 1e-6, -1e-6, 100.5, -2047.9, 3.0e-3, 16000.0])
    buf = torch.zeros(2, 5, 7, 160, device=DEV)
    buf[..., 32:128] = x.to(DEV)
    sp = ops.split_copy(buf[..., 32:128])
    raw = sp.t.cpu().numpy().view(np.float16).reshape(2, 5, 7, 3, 2, 32)       # [chunk][x0 | x1][32]
    x0, x1 = _split_np(x.numpy().reshape(2, 5, 7, 3, 32))
    assert np.array_equal(raw[..., 0, :].view(np.uint16), x0.view(np.uint16))
    assert np.array_equal(raw[..., 1, :].view(np.uint16), x1.view(np.uint16))
    back = sp.float().cpu()
    assert (back - x).abs().max() <= 2.0 ** -22 * x.abs().max()
    # with an activation on the way, and slicing at chunk boundaries only
    sp2 = ops.split_copy(buf[..., 32:128], act=3)
    close(sp2.float().cpu(), torch.tanh(x), rtol=1e-6, what="tanh + split")
    assert sp[..., 32:].shape[3] == 64
    from focusflow_official_amd._hip import FocusFlowHipError
    with pytest.raises(FocusFlowHipError):
        sp[..., 16:]


def _pack(ops, wt, cin, fmt=1):
    cout, _, kh, kw = wt.shape
    wp = torch.empty(cout, kh * kw * cin, device=DEV)
    ops.pack_conv_weight(wt.to(DEV), wp, cin)
    return ops.pack_split(wp) if fmt else wp


def _views(xs, extra=32):
    """fp32 NHWC inputs as channel slices of wider buffers (ld != C), plus their split-pair twins in buffers of another ld."""
    plain, split = [], []
    for x in xs:
        b, c, h, w = x.shape
        buf = torch.zeros(b, h, w, c + 2 * extra, device=DEV)
        buf[..., extra:extra + c] = nhwc(x)
        plain.append(buf[..., extra:extra + c])
    return plain


DMA_CASES = [
    # (segments, cout, kh, kw, B, H, W, act)
    ([256], 192, 3, 3, 2, 24, 40, 1),           # convc2
    ([128], 64, 3, 3, 1, 19, 33, 1),            # convf2, ragged plane
    ([192, 64], 126, 3, 3, 2, 16, 24, 1),       # motion conv: two segments, Cout 126
    ([128, 128], 256, 1, 5, 2, 16, 24, 0),      # z|r horizontal
    ([128, 128], 128, 5, 1, 1, 21, 30, 0),      # q vertical, ragged
    ([128], 512, 3, 3, 1, 16, 24, 1),           # heads
    ([32, 64, 32], 96, 3, 3, 1, 9, 17, 0),      # three segments, Cout off the 64 / 128 tiles, tiny plane
    ([128, 128], 256, 1, 5, 8, 48, 64, 0),      # the headline launch shape
    ([256], 192, 3, 3, 8, 48, 64, 1),           # convc2 at the headline shape: the "all channels" layout (12 waves per block)
    ([192, 64], 126, 3, 3, 8, 48, 64, 1),       # motion conv at the headline shape: "all channels", 8 waves
    ([128], 512, 3, 3, 8, 48, 64, 1),           # heads at the headline shape: two 16-wave blocks of 256 channels per tile
    ([64], 160, 3, 3, 3, 30, 170, 0),           # "all channels" on a ragged plane (30 = 5 x 6 rows, 170 = 10.6 x 16 columns), Cout 160
]


@pytest.mark.parametrize("tile", [0, 8, 4])
@pytest.mark.parametrize("case", DMA_CASES, ids=lambda c: f"c{'+'.join(map(str, c[0]))}-o{c[1]}-k{c[2]}x{c[3]}-{c[4]}x{c[5]}x{c[6]}")
def test_dma_conv_equals_fp32_route(ops, case, tile, monkeypatch):
    """conv_dma.hip over split-pair inputs against the fp32-input kernels on the same values: every tile shape
    (FF_DMA_TILE = pixel rows per block; 0 = the dispatcher's choice), outputs in fp32, in the split-pair
    format and as the second (y2) copy."""
    segs, cout, kh, kw, b, h, w, act = case
    if tile:
        monkeypatch.setenv("FF_DMA_TILE", str(tile))
    else:
        monkeypatch.delenv("FF_DMA_TILE", raising=False)
    g = torch.Generator().manual_seed(hash(str(case)) & 0xFFFF)
    cin = sum(segs)
    xs = [torch.randn(b, c, h, w, generator=g) for c in segs]
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    bias = torch.randn(cout, generator=g).to(DEV)
    res = torch.randn(b, cout, h, w, generator=g)
    pad = (kh // 2, kw // 2)
    wp = _pack(ops, wt, cin)
    plain = _views(xs)
    split = [ops.split_copy(x) for x in plain]
    ref64 = F.conv2d(torch.cat(xs, 1).double(), wt.double(), bias.cpu().double(), padding=pad) + res.double()
    ref64 = [lambda v: v, torch.relu, torch.sigmoid, torch.tanh][2 if act == 0 else 1](ref64)
    act_res = 2 if act == 0 else 1
    old = ops.conv2d(plain, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1)
    new = ops.conv2d(split, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1)
    torch.cuda.synchronize()
    newf = ops.conv2d(split, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1, w_frag=ops.pack_frag16(wp, cout))
    assert torch.equal(newf, new), "weights in fragment order (ff_pack_frag16) against the packed rows"
    close(nchw(new), ref64, rtol=2e-5, what="dma conv vs fp64")
    close(new.cpu(), old.cpu(), rtol=2e-6, what="dma conv vs fp32-input kernel")
    # split-pair output (whole, from a channel, second copy): the same values, rounded to the format
    full = (cout + 31) // 32 * 32
    buf = torch.zeros(b, h, w, full, device=DEV)
    o_s = ops.conv2d(split, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1, out=buf[..., :cout], y_split=True)
    back = ops.split_copy(buf, to_split=False)[..., :cout]
    assert isinstance(o_s, ops.SplitT)
    assert (back - new).abs().max() <= 2.0 ** -21 * max(1.0, float(new.abs().max()))
    if cout % 64 == 0:
        buf2 = torch.zeros(b, h, w, cout, device=DEV)
        raw = ops.conv2d(split, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1, out=buf2, y_split=cout // 2)
        assert torch.equal(raw[..., :cout // 2], new[..., :cout // 2])
        whole = ops.split_copy(buf2, to_split=False)
        assert (whole[..., cout // 2:] - new[..., cout // 2:]).abs().max() <= 2.0 ** -21 * max(1.0, float(new.abs().max()))
        o1, o2 = ops.conv2d(split, wp, bias, cout, kh, kw, 1, pad, res=nhwc(res), act_res=act_res, w_fmt=1, y2_split=True)
        assert torch.equal(o1, new)
        assert torch.equal(o2.t.view(torch.int32), ops.split_copy(new).t.view(torch.int32))


@pytest.mark.parametrize("shape", [(2, 16, 24), (1, 21, 30), (8, 48, 64)])
def test_dma_conv_gru_epilogues_are_bit_identical_to_the_fp32_route(ops, shape):
    """The two GRU steps in the epilogue (FF_EP_GRU_RH / FF_EP_GRU_BLEND): conv_patch.hip's instances of them use the same
    matrix instruction and summation order as conv_dma.hip - states must come out bit for bit the same, in fp32 and (y2)
    as the split pair of those very bits."""
    b, h, w = shape
    g = torch.Generator().manual_seed(b * 7 + h)
    c = 128
    hst, mot = torch.randn(b, c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    pre_zr, pre_q = torch.randn(b, 2 * c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    for kh, kw in ((1, 5), (5, 1)):
        wzr = torch.randn(2 * c, 2 * c, kh, kw, generator=g) / (2 * c * 5) ** 0.5
        wq = torch.randn(c, 2 * c, kh, kw, generator=g) / (2 * c * 5) ** 0.5
        bzr, bq = torch.randn(2 * c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
        pzr, pq = _pack(ops, wzr, 2 * c), _pack(ops, wq, 2 * c)
        pad = (kh // 2, kw // 2)
        hp, mp = _views([hst, mot])
        hs, ms = ops.split_copy(hp), ops.split_copy(mp)
        zr_old = ops.conv2d([hp, mp], pzr, bzr, 2 * c, kh, kw, 1, pad, res=nhwc(pre_zr), act_res=2, w_fmt=1, ep_rh=hp, ep_split=c)
        q_old = ops.conv2d([zr_old[..., c:], mp], pq, bq, c, kh, kw, 1, pad, res=nhwc(pre_q), act_res=3, w_fmt=1, ep_blend=(zr_old[..., :c], hp))
        zr_new = ops.conv2d([hs, ms], pzr, bzr, 2 * c, kh, kw, 1, pad, res=nhwc(pre_zr), act_res=2, w_fmt=1, ep_rh=hp, ep_split=c, y_split=c,
                            w_frag=ops.pack_frag16(pzr, 2 * c))
        assert torch.equal(zr_new[..., :c], zr_old[..., :c]), "z"
        rh_bits = ops.split_copy(zr_old[..., c:].contiguous()).t.view(torch.int32)
        assert torch.equal(zr_new[..., c:].contiguous().view(torch.int32), rh_bits), "r * h as a split pair"
        h_new, h_split = ops.conv2d([ops.SplitT(zr_new[..., c:]), ms], pq, bq, c, kh, kw, 1, pad, res=nhwc(pre_q), act_res=3, w_fmt=1,
                                    ep_blend=(zr_new[..., :c], hp), y2_split=True)
        assert torch.equal(h_new, q_old), "new state"
        assert torch.equal(h_split.t.view(torch.int32), ops.split_copy(q_old).t.view(torch.int32)), "new state as a split pair"
        # against the definition (update.py:45-50), fp64
        x = torch.cat([hst, mot], 1).double()
        zr = torch.sigmoid(F.conv2d(x, wzr.double(), bzr.cpu().double(), padding=pad) + pre_zr.double())
        z, r = zr[:, :c], zr[:, c:]
        q = torch.tanh(F.conv2d(torch.cat([r * hst.double(), mot.double()], 1), wq.double(), bq.cpu().double(), padding=pad) + pre_q.double())
        close(nchw(h_new), (1 - z) * hst.double() + z * q, rtol=2e-5, what="GRU pass vs fp64")


@pytest.mark.parametrize("th", [0, 6, 4])
@pytest.mark.parametrize("shape", [(2, 18, 32), (1, 21, 30), (1, 7, 50), (8, 48, 64)])
def test_gru_pass_as_one_launch_is_bit_identical_to_the_two_convolutions(ops, shape, th, monkeypatch):
    """csrc/gru_pass.hip: a SepConvGRU pass (z|r convolution, r * h, q convolution, blend) as ONE launch against the two
    conv_dma.hip launches with the GRU epilogues - same arithmetic in the same order, so torch.equal on the new state (fp32
    and split-pair), both tap directions, planes that are ragged against the 6 x 16 / 4 x 16 tiles, and fp64 of the definition."""
    b, h, w = shape
    if th:
        monkeypatch.setenv("FF_GRU_PASS_TH", str(th))
    else:
        monkeypatch.delenv("FF_GRU_PASS_TH", raising=False)
    g = torch.Generator().manual_seed(b * 11 + h + w)
    c = 128
    hst, mot = torch.randn(b, c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    pre_zr, pre_q = torch.randn(b, 2 * c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    hp, mp = _views([hst, mot])
    hs, ms = ops.split_copy(hp), ops.split_copy(mp)
    for d, (kh, kw) in enumerate(((1, 5), (5, 1))):
        wzr = torch.randn(2 * c, 2 * c, kh, kw, generator=g) / (2 * c * 5) ** 0.5
        wq = torch.randn(c, 2 * c, kh, kw, generator=g) / (2 * c * 5) ** 0.5
        bzr, bq = torch.randn(2 * c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
        pzr, pq = _pack(ops, wzr, 2 * c), _pack(ops, wq, 2 * c)
        fzr, fq = ops.pack_frag16(pzr, 2 * c), ops.pack_frag16(pq, c)
        pad = (kh // 2, kw // 2)
        przr, prq = nhwc(pre_zr), nhwc(pre_q)
        zr = ops.conv2d([hs, ms], pzr, bzr, 2 * c, kh, kw, 1, pad, res=przr, act_res=2, w_fmt=1, ep_rh=hp, ep_split=c, y_split=c, w_frag=fzr)
        h2, h2s = ops.conv2d([ops.SplitT(zr[..., c:]), ms], pq, bq, c, kh, kw, 1, pad, res=prq, act_res=3, w_fmt=1, ep_blend=(zr[..., :c], hp),
                             y2_split=True, w_frag=fq)
        h1, h1s = ops.gru_pass(d, hs, ms, hp, przr, prq, fzr, fq, bzr, bq, 1)
        torch.cuda.synchronize()
        assert torch.equal(h1, h2), f"new state, pass {d + 1}"
        assert torch.equal(h1s.t.view(torch.int32), h2s.t.view(torch.int32)), f"new state as a split pair, pass {d + 1}"
        x = torch.cat([hst, mot], 1).double()
        zrd = torch.sigmoid(F.conv2d(x, wzr.double(), bzr.cpu().double(), padding=pad) + pre_zr.double())
        z, r = zrd[:, :c], zrd[:, c:]
        q = torch.tanh(F.conv2d(torch.cat([r * hst.double(), mot.double()], 1), wq.double(), bq.cpu().double(), padding=pad) + pre_q.double())
        close(nchw(h1), (1 - z) * hst.double() + z * q, rtol=2e-5, what=f"GRU pass {d + 1} vs fp64")


def test_motion_tail_and_split_outputs_of_the_fp32_input_kernels(ops):
    """FF_EP_MOTION_TAIL (the motion encoder's last convolution writes torch.cat([out, flow])'s flow channels itself) and
    the split-pair epilogue of the im2col kernel (convc1 1x1 over the lookup's fp32 output, convf1 7x7 over the flow)."""
    g = torch.Generator().manual_seed(11)
    b, h, w = 2, 18, 26
    cor, flo = torch.randn(b, 192, h, w, generator=g), torch.randn(b, 64, h, w, generator=g)
    wt = torch.randn(126, 256, 3, 3, generator=g) / 48
    bias = torch.randn(126, generator=g).to(DEV)
    coords = (torch.randn(b, h, w, 2, generator=g) * 5 + 10).to(DEV).contiguous()
    wp = _pack(ops, wt, 256)
    cp, fp = _views([cor, flo])
    full = torch.zeros(b, h, w, 128, device=DEV)
    ops.conv2d([ops.split_copy(cp), ops.split_copy(fp)], wp, bias, 126, 3, 3, 1, (1, 1), act=1, w_fmt=1, out=full[..., :126], y_split=True,
               ep_motion_tail=coords)
    got = ops.split_copy(full, to_split=False)
    ref = torch.relu(F.conv2d(torch.cat([cor, flo], 1), wt, bias.cpu(), padding=1))
    close(nchw(got[..., :126]), ref, rtol=2e-5, what="motion conv")
    ys, xs_ = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    flow = coords.cpu() - torch.stack([xs_, ys], -1).float()
    x0, x1 = _split_np(flow.numpy())
    want = (x0.astype(np.float32) + x1.astype(np.float32)) * 0.25
    assert np.array_equal(got[..., 126:].cpu().numpy(), want), "flow channels = split pair of coords1 - grid"
    # im2col kernel, fp32 in -> split-pair out
    corr = torch.randn(b, 352, h, w, generator=g)
    corr[:, 324:] = 0
    w1 = torch.randn(256, 352, 1, 1, generator=g) / 18
    b1 = torch.randn(256, generator=g).to(DEV)
    p1 = _pack(ops, w1, 352)
    plain = ops.conv2d([nhwc(corr)], p1, b1, 256, 1, 1, 1, (0, 0), act=1, w_fmt=1)
    sp = ops.conv2d([nhwc(corr)], p1, b1, 256, 1, 1, 1, (0, 0), act=1, w_fmt=1, y_split=True)
    assert torch.equal(sp.t.view(torch.int32), ops.split_copy(plain).t.view(torch.int32))
    flow4 = torch.zeros(b, h, w, 4, device=DEV)
    flow4[..., :2] = torch.randn(b, h, w, 2, generator=g).to(DEV)
    w7 = torch.zeros(128, 4, 7, 7)
    w7[:, :2] = torch.randn(128, 2, 7, 7, generator=g) / 10
    p7 = _pack(ops, w7, 4)
    b7 = torch.randn(128, generator=g).to(DEV)
    plain = ops.conv2d([flow4], p7, b7, 128, 7, 7, 1, (3, 3), act=1, w_fmt=1)
    sp = ops.conv2d([flow4], p7, b7, 128, 7, 7, 1, (3, 3), act=1, w_fmt=1, y_split=True)
    assert torch.equal(sp.t.view(torch.int32), ops.split_copy(plain).t.view(torch.int32))


def test_split_inputs_fail_loudly_where_no_kernel_reads_them(ops):
    from focusflow_official_amd._hip import FocusFlowHipError
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 64, 8, 16, generator=g)
    xs = ops.split_copy(nhwc(x))
    w1 = _pack(ops, torch.randn(64, 64, 1, 1, generator=g), 64)
    with pytest.raises(FocusFlowHipError, match="split-pair"):
        ops.conv2d([xs], w1, None, 64, 1, 1, 1, (0, 0), w_fmt=1)                      # 1x1: not a conv_dma shape
    w3 = _pack(ops, torch.randn(64, 64, 3, 3, generator=g), 64)
    with pytest.raises(FocusFlowHipError, match="split-pair"):
        ops.conv2d([xs], w3, None, 64, 3, 3, 2, (1, 1), w_fmt=1)                      # stride 2
    with pytest.raises(FocusFlowHipError, match="split-pair"):
        ops.conv2d([xs, nhwc(x)], _pack(ops, torch.randn(64, 128, 3, 3, generator=g), 128), None, 64, 3, 3, 1, (1, 1), w_fmt=1)   # mixed formats


def test_split_outputs_need_room_for_whole_chunks(ops):
    """A split-pair output row holds whole 32-channel chunks: 48 channels in a 48-wide buffer would put the x1 half of the
    second chunk into the next pixel - the C ABI refuses it, ops.conv2d allocates the padded buffer."""
    from focusflow_official_amd._hip import FocusFlowHipError
    g = torch.Generator().manual_seed(3)
    x = nhwc(torch.randn(1, 64, 8, 16, generator=g))
    w = _pack(ops, torch.randn(48, 64, 3, 3, generator=g) / 24, 64)
    tight = torch.empty(1, 8, 16, 48, device=DEV)
    with pytest.raises(FocusFlowHipError, match="rounded up to 32"):
        ops.conv2d([x], w, None, 48, 3, 3, 1, (1, 1), w_fmt=1, y_split=True, out=tight)
    xs = ops.split_copy(x)
    with pytest.raises(FocusFlowHipError, match="rounded up to 32"):
        ops.conv2d([xs], w, None, 48, 3, 3, 1, (1, 1), w_fmt=1, y_split=True, out=tight)
    sp = ops.conv2d([x], w, None, 48, 3, 3, 1, (1, 1), w_fmt=1, y_split=True)          # allocates 64 channels per pixel
    assert sp.t.stride(2) == 64
    plain = ops.conv2d([x], w, None, 48, 3, 3, 1, (1, 1), w_fmt=1)
    wide = torch.zeros(1, 8, 16, 64, device=DEV)
    wide[..., :48] = plain
    got = ops.split_copy(sp.t.as_strided((1, 8, 16, 64), (8 * 16 * 64, 16 * 64, 64, 1)), to_split=False)
    close(got[..., :48].cpu(), plain.cpu(), rtol=0, atol=1e-5, what="48-channel split-pair output")


@pytest.mark.parametrize("size", [(1, 128, 160, 4), (2, 136, 200, 3), (1, 384, 512, 12)])
def test_forward_with_and_without_split_activations(det_sd, size, monkeypatch):
    """The whole forward: split-pair activations in the update block on (default) and off give the same flow; both sit
    within the 1e-3 px of the oracle.  (The GRU convolutions are bit-identical on the two routes, the plain ones differ in
    the summation order inside a 32-channel chunk: 32x32x16 against 16x16x32 MFMAs.)"""
    from focusflow_official_amd import update_block
    b, h, w, iters = size
    inp = [t.to(DEV) for t in orc.shifted_pair(b, h, w, seed=31)]
    m = _model(det_sd)
    assert update_block._SPLIT_ACT
    with torch.no_grad():
        lo_s, up_s = m(*inp, raft_iters=iters, test_mode=True)
        monkeypatch.setattr(update_block, "_SPLIT_ACT", False)
        lo_p, up_p = m(*inp, raft_iters=iters, test_mode=True)
        ref_lo, ref_up = orc.ffraft_forward(det_sd, *[t.cpu() for t in inp], raft_iters=iters, test_mode=True)
    # (12 iterations at 384 x 512 amplify summation-order noise to 3e-4 px - the fp32 route itself sits 4e-4 px from the oracle)
    close(up_s.cpu(), up_p.cpu(), rtol=0, atol=1e-4 if iters < 12 else 5e-4, what="split-pair activations on vs off")
    close(up_s.cpu(), ref_up, rtol=0, atol=1e-3, what="split-pair route vs oracle")
    close(up_p.cpu(), ref_up, rtol=0, atol=1e-3, what="fp32 route vs oracle")
    close(lo_s.cpu(), ref_lo, rtol=0, atol=1e-3, what="flow_low vs oracle")


@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 64), (1, 21, 19, 96, 96), (3, 6, 16, 128, 128)])
def test_patch_kernel_writes_split_pair_outputs(ops, shape):
    """conv_patch.hip (fp32 in, 3x3 stride 1) writes FF_FMT_SPLIT from its 32x32x16 epilogue - the first convolution of a
    residual block of the eval-BatchNorm encoder: the bytes are the split pair of what the fp32 output holds, with folded
    BatchNorm scale / shift, activation and residual in the epilogue."""
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias, sc, sh = (torch.randn(cout, generator=g).to(DEV) for _ in range(3))
    res = nhwc(torch.randn(b, cout, h, w, generator=g))
    wp = _pack(ops, wt, cin)
    kw = dict(act=1, ch_scale=sc, ch_shift=sh, res=res, act_res=1, w_fmt=1)
    plain = ops.conv2d([nhwc(x)], wp, bias, cout, 3, 3, 1, (1, 1), **kw)
    sp = ops.conv2d([nhwc(x)], wp, bias, cout, 3, 3, 1, (1, 1), y_split=True, **kw)
    assert torch.equal(sp.t.view(torch.int32), ops.split_copy(plain).t.view(torch.int32))
    ref = torch.relu(torch.relu(F.conv2d(x, wt, bias.cpu(), padding=1) * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None]) + nchw(res))
    close(nchw(plain), ref, rtol=2e-5, what="conv + folded BatchNorm + residual")


def _fusion_ref(v_img, v_mask, wa, ba, wb, bb):
    """parallel_fusion.py:98-150, '1x1conv': img' = img + conv(mask), mask' = mask + conv(img), in fp64 (NCHW in / out)."""
    img, mask = v_img.double(), v_mask.double()
    return (img + F.conv2d(mask, wa.double(), ba.double()), mask + F.conv2d(img, wb.double(), bb.double()))


@pytest.mark.parametrize("lazy", ["plain", "stem", "stage"])
@pytest.mark.parametrize("shape", [(2, 16, 24, 64), (1, 8, 16, 64), (3, 32, 12, 64), (2, 8, 12, 96), (1, 24, 20, 96)])
def test_fusion_unit_as_one_launch(ops, shape, lazy):
    """ff_fusion_pair_fwd (csrc/fusion_pair.hip): a bidirectional '1x1conv' fusion unit in one launch - against fp64, and
    against the generic route (ops.conv2d over [img, mask] with the anti-diagonal weight) it replaces; with LAZY inputs the
    loader evaluates the normalisation pass in front of the unit itself: the values it uses are ff_norm_apply's, bit for
    bit (the residual part of the output proves it: conv weights zero -> output == norm_apply's output exactly)."""
    b, h, w, c = shape
    g = torch.Generator().manual_seed(h * 7 + b)
    t = [torch.randn(b, c, h, w, generator=g) * 2 + 0.3 for _ in range(2)]
    xr = [torch.randn(b, c, h, w, generator=g) for _ in range(2)]
    wa, wb = (torch.randn(c, c, 1, 1, generator=g) / 8 for _ in range(2))
    ba, bb = (torch.randn(c, generator=g) for _ in range(2))
    td = [nhwc(v) for v in t]
    xd = [nhwc(v) for v in xr]
    if lazy == "plain":
        ins = [td[0].clone(), td[1].clone()]
        vals = [t[0], t[1]]
    else:
        res = xd if lazy == "stage" else [None, None]
        stats = [ops.norm_stats(v, per_sample=True) for v in td]
        vals = [nchw(ops.norm_apply(td[i].clone(), stats[i], True, 1e-5, act=1, res=res[i])).cpu() for i in range(2)]
        ins = [ops.LazyAct(td[i].clone(), stats[i], h * w, 1e-5, 1, res[i]) for i in range(2)]
    packs = []
    for wt, bias in ((wa, ba), (wb, bb)):
        rows = _pack(ops, wt, c)
        packs.append((ops.pack_frag16(rows, c), bias.to(DEV), rows))
    img_o, mask_o = ops.fusion_pair(ins[0], ins[1], (packs[0][0], packs[1][0]), (packs[0][1], packs[1][1]), 1)
    ref_i, ref_m = _fusion_ref(vals[0], vals[1], wa, ba, wb, bb)
    close(nchw(img_o), ref_i, rtol=2e-5, what=f"img' ({lazy})")
    close(nchw(mask_o), ref_m, rtol=2e-5, what=f"mask' ({lazy})")
    # zero weights and biases: the output is the value the unit read - norm_apply's result, exactly
    zf = torch.zeros_like(packs[0][0])
    zb = torch.zeros(c, device=DEV)
    if lazy != "plain":
        ins = [ops.LazyAct(td[i].clone(), stats[i], h * w, 1e-5, 1, res[i]) for i in range(2)]
    else:
        ins = [td[0].clone(), td[1].clone()]
    i0, m0 = ops.fusion_pair(ins[0], ins[1], (zf, zf), (zb, zb), 1)
    assert torch.equal(nchw(i0).cpu(), vals[0]) and torch.equal(nchw(m0).cpu(), vals[1])


def test_fusion_unit_kernel_refuses_what_it_cannot_read(ops):
    from focusflow_official_amd._hip import FocusFlowHipError
    g = torch.Generator().manual_seed(1)
    x = nhwc(torch.randn(1, 64, 8, 16, generator=g))
    rows = _pack(ops, torch.randn(64, 64, 1, 1, generator=g), 64)
    fr = ops.pack_frag16(rows, 64)
    bias = torch.zeros(64, device=DEV)
    with pytest.raises(FocusFlowHipError, match="multiple of"):
        ops.fusion_pair(x[:, :3].contiguous(), x[:, :3].contiguous(), (fr, fr), (bias, bias), 1)       # 48 pixels per image
    wide = torch.zeros(1, 8, 16, 128, device=DEV)
    with pytest.raises(FocusFlowHipError, match="contiguous"):
        ops.fusion_pair(wide[..., :64], wide[..., 64:], (fr, fr), (bias, bias), 1)                      # channel slices
    x128 = torch.zeros(1, 8, 16, 128, device=DEV)
    with pytest.raises(FocusFlowHipError, match="C = 128"):
        ops.fusion_pair(x128, x128.clone(), (fr, fr), (bias, bias), 1)


@pytest.mark.parametrize("size", [(2, 128, 144, 3), (1, 384, 512, 12)])
def test_forward_with_and_without_the_fusion_unit_kernel(det_sd, size, monkeypatch):
    """The whole forward with fusion units 1 - 3 of both encoders as ff_fusion_pair_fwd launches (lazy normalised inputs in
    the feature encoder) against the generic route, and against the oracle."""
    from focusflow_official_amd import cce
    b, h, w, iters = size
    inp = [t.to(DEV) for t in orc.shifted_pair(b, h, w, seed=17)]
    m = _model(det_sd)
    assert cce._FUSION_PAIR
    with torch.no_grad():
        lo_n, up_n = m(*inp, raft_iters=iters, test_mode=True)
        monkeypatch.setattr(cce, "_FUSION_PAIR", False)
        lo_o, up_o = m(*inp, raft_iters=iters, test_mode=True)
        ref_lo, ref_up = orc.ffraft_forward(det_sd, *[t.cpu() for t in inp], raft_iters=iters, test_mode=True)
    close(up_n.cpu(), up_o.cpu(), rtol=0, atol=1e-4 if iters < 12 else 5e-4, what="fusion-unit kernel on vs off")
    close(up_n.cpu(), ref_up, rtol=0, atol=1e-3, what="fusion-unit kernel vs oracle")
    close(lo_n.cpu(), ref_lo, rtol=0, atol=1e-3, what="flow_low vs oracle")


def test_fusion_unit_falls_back_when_the_kernel_cannot_take_the_shape(ops):
    """FusionUnit.run with lazy inputs on a plane that is not whole tiles (120 pixels per image): the inputs are materialised
    (ff_norm_apply) and the generic 1x1 route runs - same result as materialising by hand."""
    from focusflow_official_amd import cce
    from argparse import Namespace
    g = torch.Generator().manual_seed(9)
    b, h, w, c = 2, 10, 12, 64
    unit = cce.FusionUnit(c, "1x1conv", True).to(DEV)
    t = [nhwc(torch.randn(b, c, h, w, generator=g)) for _ in range(2)]
    xr = [nhwc(torch.randn(b, c, h, w, generator=g)) for _ in range(2)]
    st = [ops.norm_stats(v, per_sample=True) for v in t]
    with torch.no_grad():
        assert not unit._pair_kernel_ok(t[0])
        lazy = [ops.LazyAct(t[i].clone(), st[i], h * w, 1e-5, 1, xr[i]) for i in range(2)]
        m1, i1 = unit.run(lazy[1], lazy[0])
        plain = [ops.norm_apply(t[i].clone(), st[i], True, 1e-5, act=1, res=xr[i]) for i in range(2)]
        m2, i2 = unit.run(plain[1], plain[0])
    assert torch.equal(i1, i2) and torch.equal(m1, m2)
