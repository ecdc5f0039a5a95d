"""CPU tests of the host code either side of the hot path: on-disk flow formats, the 16-bit PNG
codec, augmentation primitives, dataset indexes on synthetic directory trees and the validation harness."""
import os
import struct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from focusflow_official_amd import augmentor, datasets, evaluate, frame_utils


# ---------------------------------------------------------------------------------------------------------------
# formats
# ---------------------------------------------------------------------------------------------------------------
def test_flo_known_bytes_and_roundtrip(tmp_path):
    flow = np.arange(12, dtype=np.float32).reshape(2, 3, 2) * 0.5 - 1.0      # H=2, W=3
    fn = str(tmp_path / "a.flo")
    frame_utils.writeFlow(fn, flow)
    blob = open(fn, "rb").read()
    assert blob[:4] == b"PIEH"                                               # 202021.25 as float32
    assert struct.unpack("<ii", blob[4:12]) == (3, 2)                        # width first, then height
    assert np.array_equal(np.frombuffer(blob[12:], np.float32), flow.reshape(-1))   # u,v interleaved, row-major
    back = frame_utils.readFlow(fn)
    assert back.shape == (2, 3, 2) and np.array_equal(back, flow)
    fn2 = str(tmp_path / "b.flo")
    frame_utils.writeFlow(fn2, flow[..., 0], flow[..., 1])                   # separate planes
    assert open(fn2, "rb").read() == blob
    open(fn2, "wb").write(b"XXXX" + blob[4:])
    assert frame_utils.readFlow(fn2) is None                                 # wrong magic -> None, as the reference
    assert np.array_equal(frame_utils.read_gen(fn), flow)


def test_pfm_reader(tmp_path):
    img = np.arange(24, dtype=np.float32).reshape(2, 4, 3)
    fn = str(tmp_path / "a.pfm")
    with open(fn, "wb") as f:
        f.write(b"PF\n4 2\n-1.0\n")
        np.flipud(img).astype("<f4").tofile(f)                               # rows are stored bottom-up
    assert np.array_equal(frame_utils.readPFM(fn), img)
    assert np.array_equal(frame_utils.read_gen(fn), img[:, :, :-1])          # colour PFM -> first two channels
    with open(fn, "wb") as f:
        f.write(b"Pf\n4 2\n1.0\n")
        np.flipud(img[..., 0]).astype(">f4").tofile(f)                       # positive scale = big-endian
    assert np.array_equal(frame_utils.readPFM(fn), img[..., 0])
    open(fn, "wb").write(b"P6\n")
    with pytest.raises(Exception):
        frame_utils.readPFM(fn)


def test_png16_codec_against_pil_and_itself(tmp_path):
    from PIL import Image
    g = np.random.default_rng(0)
    rgb16 = g.integers(0, 65536, (7, 5, 3)).astype(np.uint16)
    fn = str(tmp_path / "c.png")
    frame_utils.write_png16(fn, rgb16)
    assert np.array_equal(frame_utils.read_png16(fn), rgb16)
    gray16 = g.integers(0, 65536, (6, 9)).astype(np.uint16)
    frame_utils.write_png16(fn, gray16)
    assert np.array_equal(frame_utils.read_png16(fn), gray16)
    assert np.array_equal(np.array(Image.open(fn)).astype(np.uint16), gray16)     # PIL reads our 16-bit gray file
    # PIL-written files use adaptive row filters (Sub/Up/Average/Paeth): smooth content exercises all of them
    yy, xx = np.mgrid[0:33, 0:47]
    rgb8 = np.stack([(yy * 3 + xx) % 256, (yy * xx) % 256, (xx * 5) % 256], -1).astype(np.uint8)
    Image.fromarray(rgb8).save(fn)
    assert np.array_equal(frame_utils.read_png16(fn), rgb8)
    Image.fromarray((gray16 // 3).astype(np.uint16)).save(fn)
    assert np.array_equal(frame_utils.read_png16(fn), gray16 // 3)


def test_kitti_flow_roundtrip_and_layout(tmp_path):
    g = np.random.default_rng(1)
    flow = np.round(g.uniform(-300, 300, (5, 8, 2)) * 64) / 64              # representable: 1/64 px steps
    fn = str(tmp_path / "k.png")
    frame_utils.writeFlowKITTI(fn, flow)
    raw = frame_utils.read_png16(fn)
    assert raw.dtype == np.uint16 and raw.shape == (5, 8, 3)
    assert np.array_equal(raw[..., 0], (64 * flow[..., 0] + 2 ** 15).astype(np.uint16))   # R = u, G = v, B = valid
    assert np.all(raw[..., 2] == 1)
    back, valid = frame_utils.readFlowKITTI(fn)
    assert np.array_equal(back, flow.astype(np.float32)) and np.all(valid == 1)
    disp = (g.uniform(0, 200, (4, 6)) * 256).astype(np.uint16)
    disp[0, 0] = 0
    frame_utils.write_png16(fn, disp)
    f2, v2 = frame_utils.readDispKITTI(fn)
    assert np.allclose(f2[..., 0], -(disp.astype(np.float64) / 256.0)) and np.all(f2[..., 1] == 0) and not v2[0, 0] and v2[1, 1]


# ---------------------------------------------------------------------------------------------------------------
# augmentation primitives
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fx,fy", [(1.7, 1.3), (0.6, 0.8), (1.0, 1.0)])
def test_resize_linear_matches_half_pixel_bilinear(fx, fy):
    g = np.random.default_rng(2)
    img = g.uniform(-5, 5, (13, 17, 2)).astype(np.float32)
    out = augmentor.resize_linear(img, fx, fy)
    oh, ow = int(round(13 * fy)), int(round(17 * fx))
    assert out.shape == (oh, ow, 2)
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode="bilinear", align_corners=False)
    assert np.allclose(out, ref[0].permute(1, 2, 0).numpy(), atol=1e-5)
    u8 = g.integers(0, 256, (9, 11)).astype(np.uint8)
    assert augmentor.resize_linear(u8, fx, fy).dtype == np.uint8


def test_color_jitter_semantics():
    g = np.random.default_rng(3)
    img = g.integers(0, 256, (8, 10, 3)).astype(np.uint8)
    assert np.array_equal(augmentor.ColorJitter()(img), img)                 # all ranges zero -> identity
    h, s, v = augmentor._rgb_to_hsv(img / 255.0)
    assert np.allclose(augmentor._hsv_to_rgb(h, s, v), img / 255.0, atol=1e-12)
    np.random.seed(0)
    out = augmentor.ColorJitter(0.4, 0.4, 0.4, 0.5 / 3.14)(img)
    assert out.dtype == np.uint8 and out.shape == img.shape and not np.array_equal(out, img)


def test_augmentors_shapes_and_flow_consistency():
    np.random.seed(4)
    img1 = np.random.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    img2 = np.random.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    mask1 = (np.random.rand(120, 160, 1) < 0.02).astype(np.uint8) * 255
    mask2 = (np.random.rand(120, 160, 1) < 0.02).astype(np.uint8) * 255
    flow = np.ones((120, 160, 2), np.float32) * [2.0, -1.0]
    aug = augmentor.FlowAugmentor(crop_size=(64, 96), min_scale=-0.1, max_scale=0.5)
    for _ in range(8):
        a1, a2, fl, m1, m2 = aug(img1, img2, flow, mask1, mask2)
        assert a1.shape == (64, 96, 3) and a2.shape == (64, 96, 3) and fl.shape == (64, 96, 2)
        assert m1.shape == (64, 96, 1) and m2.shape == (64, 96, 1) and a1.dtype == np.uint8
        # a constant flow stays constant per channel: scaled by the resize factor, sign-flipped by a flip
        assert np.ptp(fl[..., 0]) < 1e-4 and np.ptp(fl[..., 1]) < 1e-4
    # pure flip: probabilities forced
    aug.spatial_aug_prob, aug.h_flip_prob, aug.v_flip_prob = 0.0, 1.0, 0.0
    _, _, fl, _, _ = aug(img1, img2, flow, mask1, mask2)
    assert np.allclose(fl[0, 0], [-2.0, -1.0])
    # sparse: valid vectors are scattered, everything else stays invalid
    valid = np.zeros((120, 160), np.float32)
    valid[::7, ::5] = 1
    saug = augmentor.SparseFlowAugmentor(crop_size=(64, 96))
    a1, a2, fl, va, m1, m2 = saug(img1, img2, flow * valid[..., None], valid, mask1, mask2)
    assert fl.shape == (64, 96, 2) and va.shape == (64, 96) and m1.shape == (64, 96, 1)
    assert np.all(fl[va == 0] == 0) and (va == 1).sum() > 0
    f2, v2 = augmentor.SparseFlowAugmentor.resize_sparse_flow_map(flow * valid[..., None], valid, 2.0, 2.0)
    assert f2.shape == (240, 320, 2) and v2[14, 10] == 1 and np.allclose(f2[14, 10], [4.0, -2.0]) and v2.sum() <= valid.sum()


# ---------------------------------------------------------------------------------------------------------------
# dataset indexes on synthetic trees
# ---------------------------------------------------------------------------------------------------------------
def _img(path, h=24, w=32, seed=0, mode="RGB"):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    g = np.random.default_rng(seed)
    arr = g.integers(0, 256, (h, w, 3) if mode == "RGB" else (h, w)).astype(np.uint8)
    Image.fromarray(arr).save(path)
    return arr


def _flo(path, h=24, w=32, seed=0):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    fl = np.random.default_rng(seed).uniform(-3, 3, (h, w, 2)).astype(np.float32)
    frame_utils.writeFlow(path, fl)
    return fl


def test_flying_chairs_index_and_sample(tmp_path):
    root, mroot = str(tmp_path / "chairs"), str(tmp_path / "mask")
    flows = []
    for i in range(4):
        _img(f"{root}/data/{i:05d}_img1.ppm", seed=2 * i)
        _img(f"{root}/data/{i:05d}_img2.ppm", seed=2 * i + 1)
        flows.append(_flo(f"{root}/data/{i:05d}_flow.flo", seed=i))
        _img(f"{mroot}/orb/{i:05d}_img1.png", seed=100 + i, mode="L")
        _img(f"{mroot}/orb/{i:05d}_img2.png", seed=200 + i, mode="L")
    np.savetxt(f"{root}/FlyingChairs_train_val.txt", np.array([1, 2, 1, 2]), fmt="%d")
    tr = datasets.FlyingChairs(root, mroot, split="training", mask_type="orb")
    va = datasets.FlyingChairs(root, mroot, split="validation", mask_type="orb")
    assert len(tr) == 2 and len(va) == 2 and va.flow_list[0].endswith("00001_flow.flo")
    i1, i2, fl, m1, m2, valid = va[0]
    assert i1.shape == (3, 24, 32) and i1.dtype == torch.float32 and m1.shape == (1, 24, 32) and valid.shape == (24, 32)
    assert torch.equal(fl, torch.from_numpy(flows[1]).permute(2, 0, 1)) and bool(valid.all())
    assert len(3 * tr) == 6
    aug = datasets.FlyingChairs(root, mroot, aug_params={"crop_size": (16, 24), "min_scale": -0.1, "max_scale": 0.3,
                                                         "do_flip": True}, split="training")
    s = aug[1]
    assert s[0].shape == (3, 16, 24) and s[2].shape == (2, 16, 24) and s[3].shape == (1, 16, 24)
    loader = torch.utils.data.DataLoader(va + tr, batch_size=2)
    assert next(iter(loader))[0].shape == (2, 3, 24, 32)


def test_sintel_and_kitti_indexes(tmp_path):
    root, mroot = str(tmp_path / "sintel"), str(tmp_path / "mask_s")
    for scene, n in (("alley", 3), ("cave", 2)):
        for i in range(n):
            _img(f"{root}/val/clean/{scene}/frame_{i:04d}.png", seed=i)
            _img(f"{mroot}/sift/val/clean/{scene}/frame_{i:04d}.png", seed=50 + i, mode="L")
        for i in range(n - 1):
            _flo(f"{root}/val/flow/{scene}/frame_{i:04d}.flo", seed=i)
    ds = datasets.MpiSintel(root, mroot, split="val", dstype="clean", mask_type="sift")
    assert len(ds) == 3 and len(ds.flow_list) == 3 and sorted(e[0] for e in ds.extra_info) == ["alley", "alley", "cave"]
    assert ds[0][2].shape == (2, 24, 32)

    kroot, kmroot = str(tmp_path / "kitti"), str(tmp_path / "mask_k")
    gt = np.round(np.random.default_rng(5).uniform(-20, 20, (24, 32, 2)) * 64) / 64
    for i in range(2):
        _img(f"{kroot}/val/image_2/{i:06d}_10.png", seed=i)
        _img(f"{kroot}/val/image_2/{i:06d}_11.png", seed=10 + i)
        _img(f"{kmroot}/orb/val/{i:06d}_10.png", seed=20 + i, mode="L")
        _img(f"{kmroot}/orb/val/{i:06d}_11.png", seed=30 + i, mode="L")
        os.makedirs(f"{kroot}/val/flow_occ", exist_ok=True)
        frame_utils.writeFlowKITTI(f"{kroot}/val/flow_occ/{i:06d}_10.png", gt)
    kd = datasets.KITTI(kroot, kmroot, split="val", mask_type="orb")
    assert len(kd) == 2 and kd.sparse and kd.extra_info[1] == ["000001_10.png"]
    i1, i2, fl, m1, m2, valid = kd[1]
    assert torch.equal(fl, torch.from_numpy(gt.astype(np.float32)).permute(2, 0, 1)) and bool((valid == 1).all())
    test = datasets.KITTI(kroot, kmroot, split="val", mask_type="orb")
    test.is_test = True
    t = test[0]
    assert len(t) == 5 and t[0].shape == (3, 24, 32) and t[4] == ["000000_10.png"]


# ---------------------------------------------------------------------------------------------------------------
# validation harness
# ---------------------------------------------------------------------------------------------------------------
class _CannedModel(torch.nn.Module):
    """Stands in for the flow network: returns a stored answer per call, in the model's calling convention."""

    def __init__(self, answers):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.answers, self.calls = list(answers), []

    def forward(self, image1, image2, mask1, mask2, raft_iters=12, flow_init=None, test_mode=False):
        assert test_mode and image1.shape[-2] % 8 == 0 and image1.shape[-1] % 8 == 0
        self.calls.append(raft_iters)
        return None, self.answers.pop(0)


def test_evaluate_metrics_dense_and_sparse():
    g = torch.Generator().manual_seed(0)
    h, w = 20, 30                                                            # not multiples of 8: padder in play
    gts = [torch.randn(1, 2, h, w, generator=g) * 4 for _ in range(3)]
    masks = [(torch.rand(1, 1, h, w, generator=g) < 0.1).float() * 255 for _ in range(3)]
    masks[2].zero_()                                                         # batch without key points is skipped
    offs = [torch.tensor([3.0, 4.0]), torch.tensor([0.0, 1.0]), torch.tensor([6.0, 8.0])]
    batches = [(torch.zeros(1, 3, h, w), torch.zeros(1, 3, h, w), gts[i], masks[i], masks[i], torch.ones(1, h, w))
               for i in range(3)]
    pad = evaluate.InputPadder((1, 3, h, w))
    answers = [pad.pad(gts[i] + offs[i].view(1, 2, 1, 1))[0] for i in range(3)]
    model = _CannedModel(answers)
    aepe, mepe = evaluate.evaluate_loader(model, batches, iters=32, pad_mode="sintel").dense()
    assert model.calls == [32, 32, 32]
    assert abs(aepe - (5 + 1 + 10) / 3) < 1e-5 and abs(mepe - (5 + 1) / 2) < 1e-5
    # sparse / KITTI: outliers need epe > 3 and epe > 5 % of |gt|
    gt = torch.zeros(1, 2, h, w)
    gt[:, 0] = 100.0
    valid = torch.ones(1, h, w)
    valid[:, :, :10] = 0
    pr = gt.clone()
    pr[:, 0, :, 10:20] += 4.0                                                # 4 px < 5 % of 100 -> not an outlier
    pr[:, 0, :, 20:] += 6.0                                                  # 6 px > 5 px           -> outlier
    pr[:, 0, :, :10] += 50.0                                                 # invalid pixels are ignored
    kpad = evaluate.InputPadder((1, 3, h, w), mode="kitti")
    m = torch.zeros(1, 1, h, w)
    m[..., 15] = 255
    model = _CannedModel([kpad.pad(pr)[0]])
    epe, f1, mepe = evaluate.evaluate_loader(model, [(torch.zeros(1, 3, h, w), torch.zeros(1, 3, h, w), gt, m, m, valid)],
                                             iters=32, pad_mode="kitti", sparse=True).sparse()
    assert abs(epe - 5.0) < 1e-5 and abs(f1 - 50.0) < 1e-4 and abs(mepe - 4.0) < 1e-5


def test_forward_interpolate_warm_start():
    """utils.py:26-54: a constant flow lands every vector on another grid point carrying the same vector; a flow
    that pushes everything out of the frame leaves nothing to interpolate from in that region."""
    from focusflow_official_amd.utils import forward_interpolate
    f = torch.zeros(2, 10, 14)
    f[0] += 3.0
    f[1] -= 1.0
    out = forward_interpolate(f)
    assert out.shape == (2, 10, 14) and out.dtype == torch.float32
    assert torch.all(out[0] == 3.0) and torch.all(out[1] == -1.0)
    g = torch.Generator().manual_seed(0)
    r = torch.randn(2, 10, 14, generator=g)
    out = forward_interpolate(r)
    assert torch.isfinite(out).all() and set(out[0].flatten().tolist()) <= set(r[0].flatten().tolist())


def test_flo_and_pfm_against_the_reference_readers_and_writers(tmp_path):
    """Bytes the reference's writeFlow produced and arrays its readFlow / readPFM returned
    (tests/golden/make_golden_io.py) vs this package's frame_utils on the same data."""
    from conftest import load_golden
    g = load_golden("io_formats")
    fn = str(tmp_path / "a.flo")
    frame_utils.writeFlow(fn, g["flow"])
    assert open(fn, "rb").read() == g["flo_bytes"].tobytes()
    frame_utils.writeFlow(fn, g["flow"][..., 0], g["flow"][..., 1])
    assert open(fn, "rb").read() == g["flo_bytes_uv"].tobytes()
    open(fn, "wb").write(g["flo_bytes"].tobytes())
    assert np.array_equal(frame_utils.readFlow(fn), g["flo_read"])
    pf = str(tmp_path / "c.pfm")
    open(pf, "wb").write(g["pfm_bytes"].tobytes())
    assert np.array_equal(frame_utils.readPFM(pf), g["pfm_read"])
    assert np.array_equal(frame_utils.read_gen(pf), g["pfm_read_gen"])
    open(pf, "wb").write(g["pfm_gray_bytes"].tobytes())
    assert np.array_equal(frame_utils.readPFM(pf), g["pfm_gray_read"])


def test_input_padder_and_forward_interpolate_against_reference():
    """core/utils/utils.py:7-54 run in the authoring container (tests/golden/make_golden_utils.py): pad amounts for
    Sintel / KITTI sizes, padded content, unpad round trip, and the warm-start interpolation of a random flow."""
    from conftest import load_golden
    from focusflow_official_amd.utils import InputPadder, forward_interpolate
    g = load_golden("utils_padder")
    gen = torch.Generator().manual_seed(3)
    for i, (h, w, mode) in enumerate([(436, 1024, "sintel"), (375, 1242, "kitti"), (370, 1226, "kitti"), (100, 180, "sintel"),
                                      (128, 192, "sintel")]):
        x = torch.randn(1, 2, h, w, generator=gen)
        assert abs(float(x.sum()) - float(g[f"case{i}_seedcheck"][0])) < 1e-3
        p = InputPadder(x.shape, mode=mode)
        y = p.pad(x)[0]
        assert [h, w, y.shape[-2], y.shape[-1]] + list(p._pad) == list(g[f"case{i}"])
        assert np.array_equal(y[0, :, :12, :12].numpy(), g[f"case{i}_corner"])
        assert torch.equal(p.unpad(y), x) and int(g[f"case{i}_unpad_ok"][0]) == 1
    f = torch.randn(2, 14, 18, generator=gen) * 2
    assert np.array_equal(f.numpy(), g["fi_in"])
    assert np.array_equal(forward_interpolate(f).numpy(), g["fi_out"])


def test_png_unfilter_native_matches_python_loop_for_every_filter_type(monkeypatch):
    """frame_utils._unfilter: the native scan-line reconstruction (ff_png_unfilter, host code in libfocusflow_hip.so)
    against the pure-Python recurrences, on a stream that uses all five PNG filter types, 8- and 16-bit RGB."""
    import numpy as np
    from focusflow_official_amd import _hip, frame_utils
    rng = np.random.default_rng(5)
    for bpp, w in ((6, 37), (3, 50), (2, 9)):
        h, stride = 23, w * bpp
        rows = []
        for y in range(h):
            rows.append(bytes([y % 5]) + rng.integers(0, 256, stride, dtype=np.uint8).tobytes())
        raw = b"".join(rows)
        native = frame_utils._unfilter(raw, h, stride, bpp)
        monkeypatch.setattr(_hip, "load", lambda: (_ for _ in ()).throw(RuntimeError("no library")))
        slow = frame_utils._unfilter(raw, h, stride, bpp)
        monkeypatch.undo()
        assert native.dtype == np.uint8 and np.array_equal(native, slow)


def test_invalidate_packed_drops_every_cache():
    """packed / split weights are cached per parameter version; writes through `.data` do not bump it, so the
    modules expose invalidate_packed() and call it from load_state_dict / train()."""
    from argparse import Namespace
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd.cce import PackedConv
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    pcs = [v for mod in m.modules() for v in vars(mod).values() if isinstance(v, PackedConv)]
    assert len(pcs) > 90
    for pc in pcs:
        pc._key = pc._dkey = ("stale",)
    assert m.invalidate_packed() >= len(pcs) and all(pc._key is None and pc._dkey is None for pc in pcs)
    for pc in pcs:
        pc._key = ("stale",)
    m.load_state_dict(m.state_dict())
    assert all(pc._key is None for pc in pcs)
    for pc in pcs:
        pc._key = ("stale",)
    m.eval()
    assert all(pc._key is None for pc in pcs)
