"""The oracle (oracle/ffraft_ref.py) against vectors produced by the reference.

CPU-only.  This is what makes the oracle a *pinned* checker: every fixture was
written by tests/golden/make_golden.py from the reference's own modules.
"""
import zlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ffraft_ref as orc

torch.set_num_threads(8)


def crc(t):
    return zlib.crc32(t.contiguous().numpy().tobytes())


CASES = {
    "fwd_rand_128x192_b2_it12": (lambda: orc.synthetic_inputs(2, 128, 192, seed=0), 12),
    "fwd_shift_128x192_b2_it12": (lambda: orc.shifted_pair(2, 128, 192, seed=1), 12),
    "fwd_shift_128x160_b1_it4_init": (lambda: orc.shifted_pair(1, 128, 160, seed=2), 4),
}


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference(name, det_sd):
    g = load_golden(name)
    make, iters = CASES[name]
    inp = make()
    assert [crc(inp[0]), crc(inp[1]), crc(inp[2])] == g["in_crc"].tolist(), "synthetic inputs drifted"
    finit = torch.from_numpy(g["flow_init"]) if "flow_init" in g else None
    taps = {}
    with torch.no_grad():
        flow_low, flow_up = orc.ffraft_forward(det_sd, *inp, raft_iters=iters, flow_init=finit,
                                               test_mode=True, taps=taps)
        preds = orc.ffraft_forward(det_sd, *inp, raft_iters=iters, flow_init=finit)
    # Same ATen kernels on the same machine class: expect (near) bit equality.
    tol = dict(rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(taps["fmap1"][:, ::4].numpy(), g["fmap1"], **tol)
    np.testing.assert_allclose(taps["fmap2"][:, ::4].numpy(), g["fmap2"], **tol)
    np.testing.assert_allclose(taps["cnet"][:, ::4].numpy(), g["cnet"], **tol)
    np.testing.assert_allclose(taps["pyramid"][3].numpy(), g["pyr3"], **tol)
    np.testing.assert_allclose(taps["pyramid"][2].numpy(), g["pyr2"], **tol)
    np.testing.assert_allclose(taps["pyramid"][1][::37].numpy(), g["pyr1_rows"], **tol)
    np.testing.assert_allclose(taps["pyramid"][0][::37].numpy(), g["pyr0_rows"], **tol)
    b, _, h8, w8 = taps["fmap1"].shape
    c0 = orc.coords_grid(b, h8, w8)
    np.testing.assert_allclose(orc.corr_lookup(taps["pyramid"], c0)[:1].numpy(), g["look0"], **tol)
    crand = torch.from_numpy(g["crand"])
    pyr_b0 = [p[: h8 * w8] for p in taps["pyramid"]]
    np.testing.assert_allclose(orc.corr_lookup(pyr_b0, crand).numpy(), g["look_rand"], **tol)
    np.testing.assert_allclose(flow_low.numpy(), g["flow_low"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(flow_up.numpy(), g["flow_up"], rtol=0, atol=1e-4)
    assert len(preds) == int(g["n_preds"][0])
    np.testing.assert_allclose(preds[0].numpy(), g["pred_first"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(preds[len(preds) // 2].numpy(), g["pred_mid"], rtol=0, atol=1e-4)
    # the explicit (index-replaying) lookup agrees with the grid_sample route
    ex = orc.corr_lookup_explicit(pyr_b0, crand)
    np.testing.assert_allclose(ex.numpy(), g["look_rand"], rtol=2e-6, atol=5e-5)


def test_update_block_internals(det_sd):
    g = load_golden("fwd_shift_128x192_b2_it12")
    inp = orc.shifted_pair(2, 128, 192, seed=1)
    taps = {}
    with torch.no_grad():
        orc.ffraft_forward(det_sd, *inp, raft_iters=1, test_mode=True, taps=taps)
    it0 = taps["iters"][0]
    np.testing.assert_allclose(it0["net"][:, ::8].numpy(), g["net1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(it0["up_mask"][:, ::16].numpy(), g["up_mask1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(it0["delta"].numpy(), g["delta1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(it0["flow_up"].numpy(), g["up1"], rtol=0, atol=2e-4)


def test_384x512_config1(det_sd):
    g = load_golden("fwd_shift_384x512_b1_it12")
    inp = orc.shifted_pair(1, 384, 512, seed=6)
    assert [crc(inp[0]), crc(inp[1]), crc(inp[2])] == g["in_crc"].tolist()
    with torch.no_grad():
        flow_low, flow_up = orc.ffraft_forward(det_sd, *inp, raft_iters=12, test_mode=True)
    np.testing.assert_allclose(flow_low.numpy(), g["flow_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(flow_up[:, :, ::4, ::4].numpy(), g["flow_up_sub"], rtol=0, atol=2e-4)


def test_concat_fusion(det_sd_concat):
    g = load_golden("fwd_concat_128x160_b1_it4")
    inp = orc.shifted_pair(1, 128, 160, seed=8)
    with torch.no_grad():
        fl, fu = orc.ffraft_forward(det_sd_concat, *inp, raft_iters=4, test_mode=True, fusion_type="concat")
    np.testing.assert_allclose(fu.numpy(), g["flow_up"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("ft", ["SA", "CA"])
def test_attention_fusion_units_and_variants(ft, det_sd_sa, det_sd_ca):
    """SA / CA (parallel_fusion.py:14-73): the unit alone on the reference's random (q, v), then the whole net."""
    from oracle.weights import det_tensor
    g = load_golden(f"fwd_{ft.lower()}_128x160_b1_it4")
    shapes = {"SA": {"conv_q.weight": (64, 128, 3, 3), "conv_v.0.weight": (64, 64, 3, 3), "s_map.0.weight": (1, 2, 3, 3)},
              "CA": {"conv_q.weight": (64, 128, 3, 3), "conv_q.bias": (64,), "conv_v.0.weight": (64, 64, 3, 3),
                     "conv_v.0.bias": (64,), "c_map.0.weight": (4, 64, 1, 1), "c_map.0.bias": (4,),
                     "c_map.2.weight": (64, 4, 1, 1), "c_map.2.bias": (64,)}}[ft]
    assert sorted(shapes) == sorted(str(k) for k in g["unit_keys"])
    sd = {"u." + k: det_tensor(f"unit_{ft}." + k, s) for k, s in shapes.items()}
    gen = torch.Generator().manual_seed(21)
    q, v = torch.randn(2, 64, 12, 20, generator=gen), torch.randn(2, 64, 12, 20, generator=gen)
    out = (orc._sa_unit if ft == "SA" else orc._ca_unit)(sd, "u", q, v)
    np.testing.assert_allclose(out.numpy(), g["unit_out"], rtol=0, atol=2e-5)
    inp = orc.shifted_pair(1, 128, 160, seed=8)
    with torch.no_grad():
        fl, fu = orc.ffraft_forward(det_sd_sa if ft == "SA" else det_sd_ca, *inp, raft_iters=4, test_mode=True, fusion_type=ft)
    np.testing.assert_allclose(fu.numpy(), g["flow_up"], rtol=0, atol=1e-4)


def test_train_step_matches_reference(det_sd):
    """Train-mode forward (BN batch statistics) + sequence L1 + autograd."""
    g = load_golden("train_shift_128x128_b2_it3")
    inp = orc.shifted_pair(2, 128, 128, seed=4)
    gen = torch.Generator().manual_seed(5)
    flow_gt = (torch.randn(2, 2, 128, 128, generator=gen) * 5).clamp(-400, 400)
    assert crc(flow_gt) == int(g["flow_gt_crc"][0])
    valid = torch.ones(2, 128, 128)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
          for k, v in det_sd.items()}
    preds = orc.ffraft_forward(sd, *inp, raft_iters=3, training=True)
    loss, _ = orc.sequence_l1(preds, flow_gt, valid)
    loss.backward()
    assert abs(loss.item() - g["loss"][0]) < 1e-4
    for key in [k for k in g if k.startswith("grad:")]:
        name = "flow_net." + key[5:]
        # norm3 and downsample.1 are one module in the reference; the gradient
        # arrives on the key the oracle reads (downsample.1)
        name = name.replace(".norm3.", ".downsample.1.")
        gk = sd[name].grad
        got = gk.flatten()[:: max(1, gk.numel() // 512)].numpy()
        np.testing.assert_allclose(got, g[key], rtol=1e-3, atol=2e-4 * max(1.0, float(np.abs(g[key]).max())))
    for key in [k for k in g if k.startswith("buf:")]:
        np.testing.assert_allclose(sd["flow_net." + key[4:]].numpy(), g[key], rtol=0, atol=1e-5)


def test_sampler_index_math_matches_reference():
    """lookup_taps' floor indices == the taps ATen actually touched."""
    g = load_golden("sampler_index")
    n_checked = 0
    for key in [k for k in g if k.startswith("xs_")]:
        h, w = map(int, key[3:].split("x"))
        xs = torch.from_numpy(g[key])
        n = xs.numel()
        coords = torch.stack([xs, torch.full((n,), float(h // 2))], 0).view(1, 2, 1, n)
        x0, y0, wx, wy = orc.lookup_taps(coords, [(h, w)])[0]
        lo, cnt, wlo = g[f"lo_{h}x{w}"], g[f"cnt_{h}x{w}"], g[f"wlo_{h}x{w}"]
        x0 = x0.numpy()
        wx = wx.numpy()
        two = cnt == 2  # both taps in bounds and both weights non-zero: unambiguous
        assert (x0[two] == lo[two]).all(), f"{h}x{w}: floor index differs from the reference sampler"
        np.testing.assert_allclose((1 - wx)[two], wlo[two], rtol=0, atol=3e-7)
        one = cnt == 1  # upper weight exactly 0, or one tap out of range
        ok = (x0[one] == lo[one]) | ((x0[one] == -1) & (lo[one] == 0)) | (wx[one] == 0)
        assert ok.all()
        none = cnt == 0
        assert ((x0[none] < -1) | (x0[none] >= w) | ((x0[none] == -1) & (wx[none] == 0))).all()
        n_checked += int(two.sum())
    assert n_checked > 10000
    plane = torch.from_numpy(g["kat_plane"])
    pts = torch.from_numpy(g["kat_pts"]).view(1, 9, 2)
    coords = pts.permute(0, 2, 1).reshape(1, 2, 1, 9)
    x0, y0, wx, wy = orc.lookup_taps(coords, [(48, 64)])[0]
    vals = []
    for i in range(9):
        acc = 0.0
        for dy, wyv in ((0, 1 - wy[i, 4]), (1, wy[i, 4])):
            for dx, wxv in ((0, 1 - wx[i, 4]), (1, wx[i, 4])):
                xx, yy = int(x0[i, 4]) + dx, int(y0[i, 4]) + dy
                if 0 <= xx < 64 and 0 <= yy < 48:
                    acc += float(plane[0, 0, yy, xx] * wxv * wyv)
        vals.append(acc)
    np.testing.assert_allclose(np.array(vals), g["kat_out"].reshape(-1), rtol=0, atol=1e-5)


def test_upsample_known_answer():
    g = load_golden("upsample")
    out = orc.upsample_flow(torch.from_numpy(g["flow"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-6)


def test_c_oracle_matches_reference_vectors(det_sd):
    """oracle/corr_oracle.c (plain C) against the reference's pyramid / lookup."""
    from oracle import corr_c
    g = load_golden("fwd_shift_128x192_b2_it12")
    inp = orc.shifted_pair(2, 128, 192, seed=1)
    taps = {}
    with torch.no_grad():
        orc.ffraft_forward(det_sd, *inp, raft_iters=1, test_mode=True, taps=taps)
    f1, f2 = taps["fmap1"][:1].numpy(), taps["fmap2"][:1].numpy()
    vol = corr_c.corr_volume(f1, f2)
    pyr = corr_c.pyramid(vol, 16, 24)
    np.testing.assert_allclose(pyr[0][::37][:11], g["pyr0_rows"][:11, 0], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(pyr[3], g["pyr3"][:384, 0], rtol=2e-5, atol=2e-4)
    # lookup on the REFERENCE-equivalent pyramid so only the sampler differs
    ref_pyr = [p[:384, 0].numpy().copy() for p in taps["pyramid"]]
    out, tp = corr_c.lookup(ref_pyr, g["crand"])
    np.testing.assert_allclose(out, g["look_rand"], rtol=2e-6, atol=5e-5)
    # the C taps equal the torch replay's taps (which are pinned to ATen's in the test above)
    sizes = [tuple(p.shape[-2:]) for p in ref_pyr]
    for lvl, (x0, y0, _, _) in enumerate(orc.lookup_taps(torch.from_numpy(g["crand"]), sizes)):
        assert (tp[:, lvl, 0] == x0.numpy()).all() and (tp[:, lvl, 1] == y0.numpy()).all()


LOSS_KINDS = {"EPELoss": {}, "CPCL": dict(kernel_size=5, sigma=1.7), "MixLoss": dict(kernel_size=5, sigma=1.7, lamda=0.8),
              "MixLoss_k1": dict(kernel_size=1, sigma=0.01, lamda=1)}


@pytest.mark.parametrize("kind", list(LOSS_KINDS))
def test_loss_oracle_matches_reference(kind):
    g = load_golden("losses")
    preds, gt, valid, mask = orc.loss_inputs()
    preds = [p.requires_grad_(True) for p in preds]
    loss, metrics = orc.sequence_loss(kind.split("_")[0], preds, gt, valid, mask, **LOSS_KINDS[kind])
    loss.backward()
    assert abs(loss.item() - g[kind + ":loss"][0]) < 1e-6 * max(1, abs(g[kind + ":loss"][0]))
    assert abs(metrics["epe"] - g[kind + ":epe"][0]) < 1e-5
    for i, p in enumerate(preds):
        np.testing.assert_allclose(p.grad[:, :, ::3, ::3].numpy(), g[f"{kind}:g{i}"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("ft", ["1x1conv", "concat"])
def test_ffpwc_restatement_matches_reference_layers(ft):
    """FF_PWCNET (ff_pwcnet.py:113-434) restated in oracle/pwc_ref.py vs vectors from the reference's own module run
    with the cost volume substituted (tests/golden/make_golden_pwc.py): state_dict keys/shapes, the five flows, the
    test_mode output, with and without the pre-resize."""
    import zlib
    from conftest import golden_spec
    from oracle import pwc_ref
    from oracle.weights import det_tensor
    spec = golden_spec(f"pwc_state_dict_spec_{ft}")
    if ft == "1x1conv":
        assert [(k, tuple(s)) for k, s, _ in spec] == [(k, tuple(s)) for k, s in pwc_ref.pwc_state_spec()]
    sd = {}
    for k, shp, _ in spec:
        t = det_tensor("pwc." + k, shp)
        if k in ("netExtractor.netOne.0.weight", "netExtractor.mask_netOne.0.weight"):
            t = t / 255.0
        if ".netSix.0." in k or k.startswith("netRefiner.netMain.12") or "netUpf" in k:
            t = t * 0.1
        sd[k] = t
    g = load_golden(f"pwc_fwd_{ft}")
    for tag, (b, h, w), seed in (("128x192", (2, 128, 192), 4), ("100x180", (1, 100, 180), 5)):
        gen = torch.Generator().manual_seed(seed)
        base = torch.rand(b, 3, h // 4 + 4, w // 4 + 4, generator=gen)
        i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
        i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
        m1 = (torch.rand(b, 1, h, w, generator=gen) < 0.02).float() * 255
        assert [zlib.crc32(t.contiguous().numpy().tobytes()) for t in (i1, i2, m1)] == list(g[f"in_crc_{tag}"])
        with torch.no_grad():
            flows = pwc_ref.ffpwc_forward(sd, i1, i2, m1, fusion_type=ft)
            full = pwc_ref.ffpwc_forward(sd, i1, i2, m1, test_mode=True, fusion_type=ft)
        for lvl, fl in enumerate(flows):
            np.testing.assert_allclose(fl.numpy(), g[f"flow{lvl + 2}_{tag}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(full.numpy(), g[f"full_{tag}"], rtol=0, atol=2e-6)
        assert float(np.abs(g[f"full_{tag}"]).max()) > 0.05


@pytest.mark.parametrize("modal", ["point", "frame", "neighborG"])
def test_wrapper_mask_modes_match_reference(modal, det_sd):
    """FF_RAFT_FUSION.forward (ff_raft.py:134-164): init_mask + [0,255] -> [-1,1] scaling + RAFT, against vectors from
    the reference's own wrapper (tests/golden/make_golden_wrapper.py; the modes that never call OpenCV)."""
    g = load_golden("wrapper_modes_128x160_b2_it3")
    inp = orc.shifted_pair(2, 128, 160, seed=17)
    assert [zlib.crc32(t.contiguous().numpy().tobytes()) for t in inp] == list(g["in_crc"])
    m1, m2 = orc.init_mask(inp[0], inp[1], inp[2], modal, dilate=31, kernel_size=31, kernel_sigma=5)
    np.testing.assert_allclose(m1.float().numpy()[:, :, ::2, ::2], g[f"{modal}_mask1"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(m2.float().numpy()[:, :, ::2, ::2], g[f"{modal}_mask2"], rtol=0, atol=2e-4)
    ref_in = tuple(2 * (t / 255.0) - 1.0 for t in (inp[0], inp[1], m1, m2))
    with torch.no_grad():
        fl, fu = orc.raft_forward(det_sd, *ref_in, iters=3, test_mode=True, prefix="flow_net.")
    np.testing.assert_allclose(fl.numpy(), g[f"{modal}_flow_low"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(fu.numpy(), g[f"{modal}_flow_up"], rtol=0, atol=1e-4)


def test_config5_544x960_it32_matches_reference_and_fp16_pyramid_definition():
    """BASELINE configs[4] (tests/golden/make_golden_c5.py): the oracle against the reference's own fp32 run at full size
    and full length, and the fp16-pyramid restatement (autocast semantics) against its definition: every level holds
    fp16-representable values, each pooled level is ATen's avg_pool2d of the ROUNDED level below, rounded again."""
    from conftest import golden_spec
    from oracle.weights import det_tensor
    g = load_golden("fwd_c5_544x960_b1_it32")
    sd = {k: det_tensor(k, s, flow_head_damp=float(g["damp"])) for k, s, _ in golden_spec()}
    inp = orc.shifted_pair(1, 544, 960, seed=3)
    assert [crc(inp[0]), crc(inp[1]), crc(inp[2])] == g["in_crc"].tolist(), "synthetic inputs drifted"
    with torch.no_grad():
        fl, fu = orc.ffraft_forward(sd, *inp, raft_iters=32, test_mode=True)
    np.testing.assert_allclose(fl.numpy(), g["flow_low_fp32"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(fu[:, :, ::4, ::4].numpy(), g["flow_up_sub_fp32"], rtol=0, atol=2e-4)
    # the recorded fp32-vs-fp64 spread is what makes the 1e-3 px bound of the GPU test meaningful
    assert g["spread_per_iter"][-1] < 5e-4 and np.abs(g["flow_low_fp64"]).max() > 1.0
    vol = torch.randn(1, 16 * 24, 16, 24, generator=torch.Generator().manual_seed(1)) * 50
    pyr = orc.corr_pyramid(vol, half=True)
    assert all(torch.equal(p, p.half().float()) for p in pyr)
    for lo, hi in zip(pyr[:-1], pyr[1:]):
        assert torch.equal(hi, torch.nn.functional.avg_pool2d(lo, 2, stride=2).half().float())
    full = orc.corr_pyramid(vol)
    assert torch.equal(pyr[0], full[0].half().float()) and (pyr[1] - full[1]).abs().max() <= 2.0 ** -10 * full[1].abs().max()


@pytest.mark.parametrize("modal", ["frame", "neighborG"])
def test_pwc_oracle_mask_modes_match_reference(modal):
    """oracle/pwc_ref.py with FF-PWC's non-'point' init_mask modes against the reference's own FF_PWCNET
    (tests/golden/make_golden_pwc_masks.py; cost volume = the oracle's, as in every FF-PWC fixture)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import pwc_ref
    from test_hip_pwc import _pwc_weights
    g = load_golden("pwc_mask_modes")
    sd = _pwc_weights()
    for tag, (b, h, w), seed in (("128x192", (1, 128, 192), 14), ("100x180", (1, 100, 180), 15)):
        gen = torch.Generator().manual_seed(seed)
        base = torch.rand(b, 3, h // 4 + 4, w // 4 + 4, generator=gen)
        i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
        i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
        m1 = (torch.rand(b, 1, h, w, generator=gen) < 0.02).float() * 255
        assert [crc(i1), crc(i2), crc(m1)] == g[f"{modal}_{tag}_in_crc"].tolist()
        with torch.no_grad():
            full = pwc_ref.ffpwc_forward(sd, i1, i2, m1, test_mode=True, mask_modal=modal, dilate=7, kernel_size=9, kernel_sigma=1.5)
            flows = pwc_ref.ffpwc_forward(sd, i1, i2, m1, mask_modal=modal, dilate=7, kernel_size=9, kernel_sigma=1.5)
        np.testing.assert_allclose(full.numpy(), g[f"{modal}_{tag}_full"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(flows[0].numpy(), g[f"{modal}_{tag}_flow2"], rtol=0, atol=2e-6)


def test_division_by_plane_size_in_five_fused_operations_is_the_true_quotient():
    """csrc/corr_lookup_dma.hip replaces the division of utils.py:61, 2 x / (n - 1), by reciprocal refinement with two
    exact remainders.  It must be THE correctly rounded quotient for every coordinate: all divisors a plane can have
    (1 .. 4096) against random, integer, half-integer, near-tie and huge numerators."""
    from oracle import corr_c
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-600, 600, 40000).astype(np.float32),
          np.arange(-300, 300, dtype=np.float32), np.arange(-300, 300, dtype=np.float32) + 0.5,
          (rng.integers(-(1 << 24), 1 << 24, 20000) * np.float32(2.0) ** rng.integers(-30, 8, 20000)).astype(np.float32),
          rng.standard_normal(20000).astype(np.float32) * np.float32(1e-30), np.float32([0.0, -0.0, 1e30, -1e30, 1e37])]      # |x| < 2^126: 2 x must not overflow
    # numerators whose quotient sits next to a rounding boundary: x = d * (m + 1/2 ulp) / 2 for random significands m
    m = rng.integers(1 << 23, 1 << 24, 20000).astype(np.float64)
    for d in list(range(1, 260)) + [383, 511, 959, 1023, 2047, 4095, 4096] + list(rng.integers(260, 4096, 60)):
        tie = (d * (m + rng.choice([-0.5, 0.5], m.size)) / 2 * 2.0 ** -20).astype(np.float32)
        bad, first = corr_c.div5_mismatches(np.concatenate(xs + [tie]), int(d))
        assert bad == 0, (d, bad, first)


@pytest.mark.parametrize("b,c,h,w", [(2, 37, 9, 11), (1, 196, 7, 16), (1, 32, 14, 32), (1, 16, 5, 6)])
def test_pwc_cost_volume_two_independent_restatements_agree(b, c, h, w):
    """The FF-PWC cost volume has no reference run to pin it (CuPy kernels: no CUDA here).  Two restatements that share
    no code - the tensor-wise one in oracle/pwc_ref.py and the loop-level transcription of correlation.py:7-102 in
    oracle/corr_oracle.c (padded NHWC rbot, ch % 9 - 4 on x, ch / 9 - 4 on y, 32 striding partial sums) - must agree."""
    from oracle import corr_c, pwc_ref
    g = torch.Generator().manual_seed(b * 1000 + c)
    one, two = torch.randn(b, c, h, w, generator=g), torch.randn(b, c, h, w, generator=g)
    a = pwc_ref.cost_volume(one, two).numpy()
    k = corr_c.pwc_costvolume_kernel(one.numpy(), two.numpy())
    assert a.shape == k.shape == (b, 81, h, w)
    np.testing.assert_allclose(k, a, rtol=0, atol=2e-6 * max(1.0, float(np.abs(a).max())))
    # displacement (0, 0) is channel 40 and nothing else: <one, two> / C
    np.testing.assert_allclose(k[:, 40], (one * two).sum(1).numpy() / c, rtol=0, atol=2e-6 * max(1.0, float(np.abs(a).max())))
    # the volume is not symmetric in its arguments: swapping them mirrors the displacement
    k2 = corr_c.pwc_costvolume_kernel(two.numpy(), one.numpy())
    assert np.abs(k2 - k).max() > 1e-3


def test_pwc_cost_volume_hand_derived_known_answers():
    """Channel order, displacement sign and zero padding of the FF-PWC cost volume, pinned by cases written down from the
    formula of correlation.py:46-98 (tests/golden/make_golden_pwc_known_answers.py: one-hot features -> one-hot displacement
    channel, all-ones -> the in-image indicator, a coordinate ramp -> the displaced coordinates) - independently of the two
    restatements, which both have to reproduce them exactly."""
    from oracle import corr_c, pwc_ref
    g = load_golden("pwc_costvolume_known")
    names = sorted({k.rsplit(".", 1)[0] for k in g})
    assert len(names) == 6
    for n in names:
        one, two, top = g[n + ".one"], g[n + ".two"], g[n + ".top"]
        a = pwc_ref.cost_volume(torch.from_numpy(one), torch.from_numpy(two)).numpy()
        k = corr_c.pwc_costvolume_kernel(one, two)
        np.testing.assert_allclose(a, top, rtol=0, atol=1e-5 * max(1.0, float(np.abs(top).max())), err_msg=n + " (pwc_ref.cost_volume)")
        np.testing.assert_allclose(k, top, rtol=0, atol=1e-5 * max(1.0, float(np.abs(top).max())), err_msg=n + " (corr_oracle.c)")
        if n.startswith("A_") and "outside" not in n:
            assert np.count_nonzero(a) == 1 and np.count_nonzero(k) == 1, n
