"""CPU oracle for FF-PWC's native component — TEST INFRASTRUCTURE.

Cost volume: PARITY UNPINNED (the rest of FF-PWC is pinned, see below).  The reference's own implementation (CuPy-compiled CUDA strings,
core/models/ff-pwcnet/PWCNet_Core/correlation.py:7-232) cannot run in this image (no cupy, no
CUDA device; its CPU path raises NotImplementedError, :320-321) and the reference holds no test
or golden vector for it.  `cost_volume` below restates the arithmetic those kernels spell out
(:34-102: channel = (p+4)*9+(o+4), s2o = ch%9-4 on x, s2p = ch/9-4 on y, zero padding by 4, mean
over C); its gradients come from autograd and agree with :104-166 / :168-232 by construction.

backwarp: restates ff_pwcnet.py:27-47 with the same ATen grid_sample call (minus `.cuda()`).
"""
import torch
import torch.nn.functional as F


def cost_volume(one: torch.Tensor, two: torch.Tensor) -> torch.Tensor:
    """NCHW (B,C,H,W) x2 -> (B,81,H,W)."""
    b, c, h, w = one.shape
    pad = F.pad(two, (4, 4, 4, 4))
    outs = []
    for p in range(-4, 5):          # y displacement (slow)
        for o in range(-4, 5):      # x displacement (fast)
            outs.append((one * pad[:, :, 4 + p:4 + p + h, 4 + o:4 + o + w]).sum(1, keepdim=True) / c)
    return torch.cat(outs, 1)


def backwarp(ten_input: torch.Tensor, ten_flow: torch.Tensor) -> torch.Tensor:
    b, _, h, w = ten_flow.shape
    hor = torch.linspace(-1.0 + (1.0 / w), 1.0 - (1.0 / w), w).view(1, 1, 1, -1).repeat(1, 1, h, 1)
    ver = torch.linspace(-1.0 + (1.0 / h), 1.0 - (1.0 / h), h).view(1, 1, -1, 1).repeat(1, 1, 1, w)
    grid = torch.cat([hor, ver], 1)
    flow = torch.cat([ten_flow[:, 0:1] / ((ten_input.shape[3] - 1.0) / 2.0),
                      ten_flow[:, 1:2] / ((ten_input.shape[2] - 1.0) / 2.0)], 1)
    inp = torch.cat([ten_input, ten_flow.new_ones(b, 1, h, w)], 1)
    out = F.grid_sample(inp, (grid + flow).permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = out[:, -1:].clone()
    mask[mask > 0.999] = 1.0
    mask[mask < 1.0] = 0.0
    return out[:, :-1] * mask


# ----------------------------------------------------------------------------
# FF_PWCNET forward (ff_pwcnet.py:113-434), functional restatement over a state_dict with the
# reference's key names.  PINNED, except for the cost-volume arithmetic: tests/golden/make_golden_pwc.py runs the
# reference's own FF_PWCNET (Extractor / Decoder / Refiner / backwarp / preprocess / test_mode resize) with two
# stand-in modules for what this image lacks - an empty cv2 and a `correlation` whose FunctionCorrelation is
# `cost_volume` above - and tests/test_oracle_golden.py holds this restatement to those vectors.
# ----------------------------------------------------------------------------
LEVEL_CH = [16, 32, 64, 96, 128, 196]
EXTRACTOR = ["netOne", "netTwo", "netThr", "netFou", "netFiv", "netSix"]
BACKWARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}      # fltBackwarp, ff_pwcnet.py:276
DECODER = {6: "netSix", 5: "netFiv", 4: "netFou", 3: "netThr", 2: "netTwo"}


def _c(sd, name, x, stride=1, padding=1, dilation=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding, dilation=dilation)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def pwc_extractor(sd, p, x, mask, fusion_type="1x1conv"):
    """ff_pwcnet.py:237-264."""
    feats = []
    for lvl, name in enumerate(EXTRACTOR):
        for branch in ("", "mask_"):
            t = x if branch == "" else mask
            t = _lrelu(_c(sd, f"{p}.{branch}{name}.0", t, 2))
            t = _lrelu(_c(sd, f"{p}.{branch}{name}.2", t))
            t = _lrelu(_c(sd, f"{p}.{branch}{name}.4", t))
            if branch == "":
                x = t
            else:
                mask = t
        fu = f"{p}.fusion{lvl + 1}"

        def fc(name, t):
            return F.conv2d(t, sd[f"{fu}.{name}.conv.weight"], sd[f"{fu}.{name}.conv.bias"])

        if fusion_type == "concat":   # parallel_fusion.py:76-84: conv1x1(cat[q, v])
            x_new = fc("mask2img", torch.cat([x, mask], 1))
            if lvl < 5:
                mask = fc("img2mask", torch.cat([mask, x], 1))
        else:                         # '1x1conv' (:87-95): q + conv1x1(v)
            x_new = x + fc("mask2img", mask)
            if lvl < 5:   # fusion6 is uni-directional
                mask = mask + fc("img2mask", x)
        x = x_new
        feats.append(x)
    return feats


def pwc_decoder(sd, p, level, one, two, prev):
    """ff_pwcnet.py:307-343."""
    if prev is None:
        feat = _lrelu(cost_volume(one, two))
        flow = None
    else:
        flow = F.conv_transpose2d(prev["tenFlow"], sd[p + ".netUpflow.weight"], sd[p + ".netUpflow.bias"], stride=2, padding=1)
        upfeat = F.conv_transpose2d(prev["tenFeat"], sd[p + ".netUpfeat.weight"], sd[p + ".netUpfeat.bias"], stride=2, padding=1)
        vol = _lrelu(cost_volume(one, backwarp(two, flow * BACKWARP_SCALE[level])))
        feat = torch.cat([vol, one, flow, upfeat], 1)
    for name in ("netOne", "netTwo", "netThr", "netFou", "netFiv"):
        feat = torch.cat([_lrelu(_c(sd, f"{p}.{name}.0", feat)), feat], 1)
    return {"tenFlow": _c(sd, p + ".netSix.0", feat), "tenFeat": feat}


def pwc_refiner(sd, p, feat):
    """ff_pwcnet.py:346-370."""
    x = feat
    dils = [1, 2, 4, 8, 16, 1, 1]
    for i, d in enumerate(dils):
        x = _c(sd, f"{p}.netMain.{2 * i}", x, 1, d, d)
        if i < 6:
            x = _lrelu(x)
    return x


def ffpwc_forward(sd, image1, image2, mask1, mask2=None, test_mode=False, mask_channel=3, fusion_type="1x1conv",
                  mask_modal="point", dilate=31, kernel_size=31, kernel_sigma=5):
    """ff_pwcnet.py:405-433, including preprocess (:391-403): sizes that are not multiples of 64 are bilinearly resized
    first; then init_mask (:61-110) on the resized inputs - the same five modes as FF-RAFT's (ffraft_ref.init_mask
    restates both: the two functions differ only in names).  Inputs stay in [0,255] (FF-PWC does not normalise)."""
    from .ffraft_ref import init_mask
    b, _, h0, w0 = image1.shape
    h, w = -(-h0 // 64) * 64, -(-w0 // 64) * 64
    image1, image2, mask1 = (F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False) for t in (image1, image2, mask1))
    m1, m2 = init_mask(image1, image2, mask1, mask_modal, mask_channel, dilate, kernel_size, kernel_sigma)
    m1, m2 = m1.to(image1.dtype), m2.to(image1.dtype)
    f1 = pwc_extractor(sd, "netExtractor", image1, m1, fusion_type)
    f2 = pwc_extractor(sd, "netExtractor", image2, m2, fusion_type)
    est = None
    flows = []
    for level in (6, 5, 4, 3, 2):
        est = pwc_decoder(sd, DECODER[level], level, f1[level - 1], f2[level - 1], est)
        if level == 2:
            est["tenFlow"] = est["tenFlow"] + pwc_refiner(sd, "netRefiner", est["tenFeat"])
        flows.insert(0, est["tenFlow"])
    if test_mode:
        out = F.interpolate(est["tenFlow"], size=(h0, w0), mode="bilinear", align_corners=False)
        out[:, 0] = out[:, 0] * w0 / w
        out[:, 1] = out[:, 1] * h0 / h
        return out
    return flows


def pwc_state_spec():
    """(key, shape) list of FF_PWCNET's state_dict (1x1conv fusion), derived from ff_pwcnet.py:123-370."""
    spec = []

    def conv(name, co, ci, k):
        spec.append((name + ".weight", (co, ci, k, k)))
        spec.append((name + ".bias", (co,)))

    def extractor_stage(prefix, cin, c):
        for i, (a, bb) in enumerate(((cin, c), (c, c), (c, c))):
            conv(f"netExtractor.{prefix}.{2 * i}", bb, a, 3)

    cin = 3
    order = []
    for lvl, name in enumerate(EXTRACTOR):
        order.append((name, cin, LEVEL_CH[lvl]))
        cin = LEVEL_CH[lvl]
    # registration order in the reference: netOne, mask_netOne, fusion1, netTwo, ...
    for lvl, (name, ci, c) in enumerate(order):
        extractor_stage(name, ci, c)
        extractor_stage("mask_" + name, ci, c)
        conv(f"netExtractor.fusion{lvl + 1}.mask2img.conv", c, c, 1)
        if lvl < 5:
            conv(f"netExtractor.fusion{lvl + 1}.img2mask.conv", c, c, 1)
    cur = {6: 81, 5: 81 + 128 + 4, 4: 81 + 96 + 4, 3: 81 + 64 + 4, 2: 81 + 32 + 4}
    for level in (2, 3, 4, 5, 6):     # self.netTwo ... self.netSix
        p = DECODER[level]
        if level < 6:
            spec.append((p + ".netUpflow.weight", (2, 2, 4, 4)))
            spec.append((p + ".netUpflow.bias", (2,)))
            spec.append((p + ".netUpfeat.weight", (cur[level + 1] + 448, 2, 4, 4)))
            spec.append((p + ".netUpfeat.bias", (2,)))
        c = cur[level]
        for name, add, co in (("netOne", 0, 128), ("netTwo", 128, 128), ("netThr", 256, 96), ("netFou", 352, 64),
                              ("netFiv", 416, 32), ("netSix", 448, 2)):
            conv(f"{p}.{name}.0", co, c + add, 3)
    chans = [(565, 128), (128, 128), (128, 128), (128, 96), (96, 64), (64, 32), (32, 2)]
    for i, (ci, co) in enumerate(chans):
        conv(f"netRefiner.netMain.{2 * i}", co, ci, 3)
    return spec
