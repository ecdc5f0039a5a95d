"""FF-PWC native component: cost volume fwd/bwd and backwarp against the CPU
oracle (oracle/pwc_ref.py).  The cost volume's parity is UNPINNED (the reference's CuPy kernels cannot
run here and have no vectors); the oracle is the definition those kernels implement."""
import numpy as np
import pytest
import torch

from oracle import pwc_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-5, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), f"{what}: max err {err:.3e}"


# FF-PWC pyramid levels for a 448x1024 input (BASELINE config 4) scaled down, plus ragged sizes
@pytest.mark.parametrize("b,c,h,w", [(1, 32, 28, 64), (2, 64, 14, 32), (1, 96, 7, 16), (1, 128, 9, 21), (1, 196, 7, 16), (2, 32, 17, 35)])
def test_cost_volume_forward_backward(b, c, h, w):
    from focusflow_official_amd import pwc
    g = torch.Generator().manual_seed(c + h)
    one = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    two = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    gy = torch.randn(b, 81, h, w, generator=g)
    ref = pwc_ref.cost_volume(one, two)
    ref.backward(gy)
    od, td = nhwc(one).requires_grad_(True), nhwc(two).requires_grad_(True)
    out = pwc.FunctionCorrelation(od, td)
    close(nchw(out), ref.detach(), what="cost volume")
    out.backward(nhwc(gy))
    close(nchw(od.grad), one.grad, what="grad one")
    close(nchw(td.grad), two.grad, what="grad two")


def test_cost_volume_hand_derived_known_answers():
    """The HIP cost volume against cases written down from the formula of correlation.py:46-98 (generating script with the
    derivation: tests/golden/make_golden_pwc_known_answers.py): one-hot features -> exactly one non-zero value in the channel
    of that displacement, all-ones -> the in-image indicator (zero padding), a coordinate ramp -> displaced coordinates
    (channel k -> k + 1 moves +1 in x, k -> k + 9 moves +1 in y).  These pin channel order, sign and borders without either
    restatement of the kernel."""
    from conftest import load_golden
    from focusflow_official_amd import pwc
    g = load_golden("pwc_costvolume_known")
    for n in sorted({k.rsplit(".", 1)[0] for k in g}):
        one, two, top = (torch.from_numpy(g[n + s]) for s in (".one", ".two", ".top"))
        c = one.shape[1]
        pad = (-c) % 4                                   # the kernels read channels in groups of four: zero channels add nothing ...
        o4 = torch.nn.functional.pad(one, (0, 0, 0, 0, 0, pad))
        t4 = torch.nn.functional.pad(two, (0, 0, 0, 0, 0, pad))
        out = nchw(pwc.FunctionCorrelation(nhwc(o4), nhwc(t4))) * ((c + pad) / c)      # ... but the mean is over the padded count
        close(out, top, tol=1e-6, what=n)
        if n.startswith("A_") and "outside" not in n:
            assert int((out != 0).sum()) == 1, n


def test_cost_volume_properties_full_size():
    """BASELINE config 4 level-2 size (112x256, C=32): displacement structure."""
    from focusflow_official_amd import pwc
    g = torch.Generator().manual_seed(0)
    one = torch.randn(1, 112, 256, 32, generator=g).to(DEV)
    out = pwc.FunctionCorrelation(one, one)
    # zero displacement = mean of squares; symmetric displacements mirror each other
    close(out[..., 40].cpu(), (one * one).mean(-1).cpu(), what="centre")
    a = out[0, 10:100, 10:200, 4 * 9 + 6]          # (p, o) = (0, +2)
    bb = out[0, 10:100, 12:202, 4 * 9 + 2]         # (0, -2) seen from the shifted pixel
    close(a.cpu(), bb.cpu(), what="mirror")
    shifted = torch.roll(one, shifts=(-3, 2), dims=(1, 2))   # two[y,x] = one[y+3, x-2]
    out2 = pwc.FunctionCorrelation(one, shifted)
    best = out2[0, 20:90, 20:230].mean(dim=(0, 1)).argmax().item()
    assert best == (-3 + 4) * 9 + (2 + 4)


@pytest.mark.parametrize("b,c,h,w,scale", [(2, 32, 28, 64, 3.0), (1, 96, 7, 16, 1.0), (1, 64, 14, 33, 40.0)])
def test_backwarp(b, c, h, w, scale):
    from focusflow_official_amd import pwc
    g = torch.Generator().manual_seed(h)
    inp = torch.randn(b, c, h, w, generator=g)
    flow = torch.randn(b, 2, h, w, generator=g) * scale
    ref = pwc_ref.backwarp(inp, flow)
    out = pwc.backwarp(nhwc(inp), nhwc(flow))
    got = nchw(out)
    # the validity mask thresholds a sum of weights at 0.999: ignore the few pixels within rounding of it
    diff = (got - ref).abs().amax(1)
    bad = (diff > 1e-4).float().mean().item()
    assert bad < 2e-3, f"{bad * 100:.3f}% of pixels differ"


def _pwc_weights():
    """Deterministic weights, tamed for unnormalised [0,255] inputs: the two stem convs absorb 1/255 and the
    flow heads are damped so warps stay inside the image."""
    from oracle.weights import det_tensor
    sd = {}
    for k, s in pwc_ref.pwc_state_spec():
        t = det_tensor("pwc." + k, s)
        if k in ("netExtractor.netOne.0.weight", "netExtractor.mask_netOne.0.weight"):
            t = t / 255.0
        if ".netSix.0." in k or k.startswith("netRefiner.netMain.12") or "netUpf" in k:
            t = t * 0.1
        sd[k] = t
    return sd


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_ffpwcnet_forward_matches_oracle(precision):
    """FF_PWCNET (ff_pwcnet.py:405-433) end to end on HIP vs the CPU restatement (parity unpinned: the
    reference module cannot run in this image)."""
    from argparse import Namespace
    from focusflow_official_amd import ops as _ops
    from focusflow_official_amd.pwcnet import FF_PWCNET
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
    sd = _pwc_weights()
    m = FF_PWCNET(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    base = torch.rand(2, 3, 36, 52, generator=g)
    i1 = torch.nn.functional.interpolate(base, size=(128, 192), mode="bilinear", align_corners=False) * 255
    i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
    m1 = (torch.rand(2, 1, 128, 192, generator=g) < 0.02).float() * 255
    with torch.no_grad():
        ref = pwc_ref.ffpwc_forward(sd, i1, i2, m1)
        ref_full = pwc_ref.ffpwc_forward(sd, i1, i2, m1, test_mode=True)
    prev = _ops.conv_precision()
    _ops.set_conv_precision(precision)
    try:
        with torch.no_grad():
            got = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV))
            got_full = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV), test_mode=True)
    finally:
        _ops.set_conv_precision(prev)
    assert len(got) == 5
    for lvl, (a, r) in enumerate(zip(got, ref)):
        assert a.shape == r.shape
        close(a.cpu(), r, tol=2e-4, what=f"flow level {lvl + 2}")
    close(got_full.cpu(), ref_full, tol=2e-4, what="test_mode flow")
    assert float(ref_full.abs().max()) > 0.05, "degenerate test: flow is ~0"
    # preprocess path (ff_pwcnet.py:391-403): 100x180 is resized to 128x192, the flow is resized / rescaled back
    c1, c2, cm = i1[..., :100, :180].contiguous(), i2[..., :100, :180].contiguous(), m1[..., :100, :180].contiguous()
    with torch.no_grad():
        ref_small = pwc_ref.ffpwc_forward(sd, c1, c2, cm, test_mode=True)
        got_small = m(c1.to(DEV), c2.to(DEV), cm.to(DEV), cm.to(DEV), test_mode=True)
        got_list = m(c1.to(DEV), c2.to(DEV), cm.to(DEV), cm.to(DEV))
    assert got_small.shape == (2, 2, 100, 180) and got_list[0].shape == (2, 2, 32, 48)
    assert (m.origin_H, m.origin_W, m.new_H, m.new_W) == (100, 180, 128, 192)
    close(got_small.cpu(), ref_small, tol=2e-4, what="test_mode flow after pre-resize")


@pytest.mark.parametrize("ft", ["1x1conv", "concat"])
def test_ffpwcnet_matches_reference_layer_vectors(ft):
    """FF_PWCNET on HIP against vectors produced by the reference's own module (cost volume substituted by the
    oracle's definition, tests/golden/make_golden_pwc.py): five flows and the test_mode output, with and without the
    pre-resize, and the state_dict keys/shapes."""
    from argparse import Namespace
    from conftest import golden_spec, load_golden
    from focusflow_official_amd.pwcnet import FF_PWCNET
    from oracle.weights import det_tensor
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE=ft))
    m = FF_PWCNET(cfg)
    spec = golden_spec(f"pwc_state_dict_spec_{ft}")
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s, _ in spec]
    sd = {}
    for k, shp, _ in spec:
        t = det_tensor("pwc." + k, shp)
        if k in ("netExtractor.netOne.0.weight", "netExtractor.mask_netOne.0.weight"):
            t = t / 255.0
        if ".netSix.0." in k or k.startswith("netRefiner.netMain.12") or "netUpf" in k:
            t = t * 0.1
        sd[k] = t
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    g = load_golden(f"pwc_fwd_{ft}")
    for tag, (b, h, w), seed in (("128x192", (2, 128, 192), 4), ("100x180", (1, 100, 180), 5)):
        gen = torch.Generator().manual_seed(seed)
        base = torch.rand(b, 3, h // 4 + 4, w // 4 + 4, generator=gen)
        i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
        i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
        m1 = (torch.rand(b, 1, h, w, generator=gen) < 0.02).float() * 255
        with torch.no_grad():
            flows = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV))
            full = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV), test_mode=True)
        for lvl, fl in enumerate(flows):
            close(fl.cpu(), torch.from_numpy(g[f"flow{lvl + 2}_{tag}"]), tol=2e-4, what=f"{ft} {tag} flow level {lvl + 2}")
        close(full.cpu(), torch.from_numpy(g[f"full_{tag}"]), tol=2e-4, what=f"{ft} {tag} test_mode flow")


@pytest.mark.parametrize("b,c,h,w,scale", [(2, 32, 28, 40, 5.0), (1, 196, 7, 16, 0.625), (1, 64, 17, 23, 2.5)])
def test_backwarp_backward(b, c, h, w, scale):
    """Gradients of backwarp w.r.t. the warped features and the flow vs autograd through the oracle's restatement
    (ATen grid_sample + the constant validity mask, ff_pwcnet.py:27-47)."""
    from focusflow_official_amd import pwc
    g = torch.Generator().manual_seed(b * 100 + c)
    x = torch.randn(b, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    fl = (torch.randn(b, 2, h, w, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    gout = torch.randn(b, c, h, w, generator=g, dtype=torch.float64)
    ref = pwc_ref.backwarp(x, fl * scale)
    ref.backward(gout)
    xd = x.detach().float().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    fd = torch.zeros(b, h, w, 4, device=DEV)
    fd[..., :2] = fl.detach().float().permute(0, 2, 3, 1).to(DEV)
    fd.requires_grad_(True)
    out = pwc.backwarp(xd, fd, scale)
    close(out.detach().cpu().permute(0, 3, 1, 2), ref.detach(), tol=2e-5, what="backwarp fwd")
    out.backward(gout.float().permute(0, 2, 3, 1).contiguous().to(DEV))
    # pixels whose sample position sits within rounding distance of a cell border may pick the neighbouring cell in
    # fp32: compare with a small budget of outliers, as the forward test does
    dx = xd.grad.cpu().permute(0, 3, 1, 2).double()
    dfl = fd.grad.cpu()[..., :2].permute(0, 3, 1, 2).double()
    for got, want, what in ((dx, x.grad, "d input"), (dfl, fl.grad, "d flow")):
        err = (got - want).abs() / (want.abs().max() + 1e-12)
        assert float((err > 2e-4).float().mean()) < 2e-3, f"{what}: {float(err.max()):.3e}"
    assert float(fd.grad[..., 2:].abs().max()) == 0.0


def test_ffpwcnet_training_step_gradients_match_oracle():
    """FF_PWCNET with gradients recorded: the five flows and a selection of parameter gradients (extractor, fusion,
    transposed convs, DenseNet convs, flow heads, dilated refiner) vs autograd through the CPU restatement."""
    from argparse import Namespace
    from focusflow_official_amd.pwcnet import FF_PWCNET
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
    sd = _pwc_weights()
    m = FF_PWCNET(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(6)
    base = torch.rand(1, 3, 36, 52, generator=g)
    i1 = torch.nn.functional.interpolate(base, size=(128, 192), mode="bilinear", align_corners=False) * 255
    i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
    m1 = (torch.rand(1, 1, 128, 192, generator=g) < 0.02).float() * 255
    weights = [0.005, 0.01, 0.02, 0.08, 0.32]            # the reference's multi-scale weighting, finest level first
    flows = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV))
    assert len(flows) == 5 and flows[0].shape == (1, 2, 32, 48)
    sum(wt * f.abs().sum() for wt, f in zip(weights, flows)).backward()
    rsd = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    ref = pwc_ref.ffpwc_forward(rsd, i1.double(), i2.double(), m1.double())
    sum(wt * f.abs().sum() for wt, f in zip(weights, ref)).backward()
    for lvl, (a, r) in enumerate(zip(flows, ref)):
        close(a.detach().cpu(), r.detach(), tol=2e-4, what=f"flow level {lvl + 2}")
    params = dict(m.named_parameters())
    names = ["netExtractor.netOne.0.weight", "netExtractor.mask_netThr.2.weight", "netExtractor.netSix.4.bias",
             "netExtractor.fusion2.mask2img.conv.weight", "netSix.netOne.0.weight", "netFiv.netUpflow.weight",
             "netFiv.netUpfeat.weight", "netFou.netThr.0.weight", "netThr.netSix.0.weight", "netTwo.netFiv.0.bias",
             "netTwo.netUpfeat.bias", "netRefiner.netMain.0.weight", "netRefiner.netMain.6.weight", "netRefiner.netMain.12.weight"]
    for n in names:
        got, want = params[n].grad, rsd[n].grad
        assert got is not None, n
        err = float((got.cpu().double() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
        assert err < 5e-3, f"{n}: relative error {err:.3e}"
    assert all(p.grad is not None for p in m.parameters())


@pytest.mark.parametrize("modal", ["frame", "neighborG", "neighborE", "context"])
def test_ffpwcnet_mask_modes(modal):
    """FF-PWC's init_mask modes other than 'point' (ff_pwcnet.py:61-110), applied to the pre-processed (resized) inputs,
    values left in [0,255].  'frame' and 'neighborG' against the REFERENCE's own FF_PWCNET (tests/golden/
    make_golden_pwc_masks.py); all four against the oracle ('neighborE' / 'context' use OpenCV's ellipse element, which is
    restated from its published algorithm: unpinned)."""
    from argparse import Namespace
    from conftest import load_golden
    from focusflow_official_amd.pwcnet import FF_PWCNET
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL=modal, KERNEL_SIZE=9, KERNEL_SIGMA=1.5, MASK_DILATE=7),
                    MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
    sd = _pwc_weights()
    m = FF_PWCNET(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    g = load_golden("pwc_mask_modes") if modal in ("frame", "neighborG") else None
    for tag, (b, h, w), seed in (("128x192", (1, 128, 192), 14), ("100x180", (1, 100, 180), 15)):
        gen = torch.Generator().manual_seed(seed)                      # = make_golden_pwc.inputs(b, h, w, seed)
        base = torch.rand(b, 3, h // 4 + 4, w // 4 + 4, generator=gen)
        i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
        i2 = torch.roll(i1, shifts=(2, -3), dims=(2, 3))
        m1 = (torch.rand(b, 1, h, w, generator=gen) < 0.02).float() * 255
        with torch.no_grad():
            got = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV))
            got_full = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV), test_mode=True)
            ref = pwc_ref.ffpwc_forward(sd, i1, i2, m1, mask_modal=modal, dilate=7, kernel_size=9, kernel_sigma=1.5)
            ref_full = pwc_ref.ffpwc_forward(sd, i1, i2, m1, test_mode=True, mask_modal=modal, dilate=7, kernel_size=9, kernel_sigma=1.5)
        close(got[0].cpu(), ref[0], tol=2e-4, what=f"{modal} {tag} finest flow vs oracle")
        close(got_full.cpu(), ref_full, tol=2e-4, what=f"{modal} {tag} test_mode flow vs oracle")
        if g is not None:
            import zlib
            crc = lambda t: zlib.crc32(t.contiguous().numpy().tobytes())  # noqa: E731
            assert [crc(i1), crc(i2), crc(m1)] == g[f"{modal}_{tag}_in_crc"].tolist(), "synthetic inputs drifted"
            close(got[0].cpu(), g[f"{modal}_{tag}_flow2"], tol=2e-4, what=f"{modal} {tag} finest flow vs reference")
            close(got_full.cpu(), g[f"{modal}_{tag}_full"], tol=2e-4, what=f"{modal} {tag} test_mode flow vs reference")


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,b,h,w", [(4, 2, 2, 7, 16), (644, 2, 1, 14, 32), (36, 1, 1, 5, 9)])
def test_direct_transposed_conv_matches_conv_transpose2d(cin, cout, b, h, w):
    """ff_deconv4x4s2_small (netUpflow / netUpfeat, ff_pwcnet.py:243-244) against nn.functional.conv_transpose2d in fp64."""
    import torch.nn.functional as F
    from focusflow_official_amd import ops
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5        # ConvTranspose2d layout [Cin][Cout][k][k]
    bias = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), bias.double(), stride=2, padding=1).float()
    wf = wt.permute(1, 0, 2, 3).flip(2, 3).contiguous().cuda()              # the equivalent forward conv's weight
    rows = torch.empty((cout, 16 * cin), device="cuda")
    ops.pack_conv_weight(wf, rows, cin, 0)
    out = torch.full((b, 2 * h, 2 * w, 4), 7.0, device="cuda")               # a wider slot: channels >= cout stay untouched
    ops.deconv4x4s2_small(x.permute(0, 2, 3, 1).contiguous().cuda(), rows, bias.cuda(), cout, out)
    got = out[..., :cout].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert bool((out[..., cout:] == 7.0).all())
