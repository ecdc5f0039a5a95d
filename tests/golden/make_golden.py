#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (authoring container only).

Imports the reference's own ``FF_RAFT_Core.{raft,corr,update,utils}`` modules
from /root/reference (read-only; nothing of it is copied), fills them with the
deterministic name-hashed weights of ``oracle/weights.py`` and records
inputs' checksums + outputs as small fixtures.  The GPU box never sees the
reference — only these vectors travel.

``FF_RAFT_Core/ff_raft.py`` (init_mask + the input scaling wrapper) is not imported
HERE: it pulls in ``cv2`` at module scope, which this image lacks.  These fixtures pin
``RAFT.forward`` (raft.py:173-236) on already-normalised inputs; the wrapper itself is
pinned separately by ``make_golden_wrapper.py`` (empty ``cv2`` stand-in, the mask modes
that never call it).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/core/models/ff-raft")
sys.dont_write_bytecode = True

from FF_RAFT_Core.raft import RAFT  # noqa: E402  (reference)
from FF_RAFT_Core.corr import CorrBlock  # noqa: E402  (reference)
from FF_RAFT_Core.utils.utils import bilinear_sampler, coords_grid  # noqa: E402  (reference)

from oracle import ffraft_ref as orc  # noqa: E402
from oracle.weights import det_tensor  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def crc(t: torch.Tensor) -> int:
    return zlib.crc32(t.contiguous().numpy().tobytes())


def cfg(fusion_type="1x1conv"):
    return Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                     MODEL=Namespace(FUSION_TYPE=fusion_type, LOAD_MODULE_TO_BRANCH=False))


def build_reference(fusion_type="1x1conv"):
    """RAFT exactly as FF_RAFT_FUSION builds it (ff_raft.py:108-110)."""
    net = RAFT(in_channels=256, small=False, dropout=0.0, alternate_corr=False, abandon_fnet=False,
               inside_fusion="parallel", fuse_cnet=True, cfg=cfg(fusion_type))
    sd = net.state_dict()
    # keys as seen through FF_RAFT_FUSION: 'flow_net.' prefix (ff_raft.py:108)
    net.load_state_dict({k: det_tensor("flow_net." + k, v.shape) for k, v in sd.items()}, strict=True)
    return net


def np32(t):
    return t.detach().float().contiguous().numpy()


def case_forward(net, name, inputs, iters, flow_init=None):
    image1, image2, mask1, mask2 = inputs
    i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
    net.eval()
    rec = {}
    with torch.no_grad():
        fmap1 = net.fnet(i1, m1).float()
        fmap2 = net.fnet(i2, m2).float()
        cnet = net.cnet(i1, m1)
        corr_fn = CorrBlock(fmap1, fmap2, radius=4)
        b, _, h8, w8 = fmap1.shape
        c0 = coords_grid(b, h8, w8, device=fmap1.device)
        look0 = corr_fn(c0)
        g = torch.Generator().manual_seed(77)
        crand = c0 + (torch.rand(c0.shape, generator=g) * 16 - 8)
        look_r = corr_fn(crand)
        flow_low, flow_up = net(i1, i2, m1, m2, iters=iters, flow_init=flow_init, test_mode=True)
        preds = net(i1, i2, m1, m2, iters=iters, flow_init=flow_init, test_mode=False)
        # one hand-rolled iteration to expose update-block internals
        net_h, inp = torch.split(cnet, [128, 128], dim=1)
        net_h, inp = torch.tanh(net_h), torch.relu(inp)
        n1, up_mask1, delta1 = net.update_block(net_h, inp, look0, c0 - c0)
        up1 = net.upsample_flow(delta1, up_mask1)
    rec.update(
        in_crc=np.array([crc(image1), crc(image2), crc(mask1)], dtype=np.int64),
        fmap1=np32(fmap1[:, ::4]), fmap2=np32(fmap2[:, ::4]), cnet=np32(cnet[:, ::4]),
        fmap_stats=np.array([fmap1.mean().item(), fmap1.std().item(), fmap2.mean().item(),
                             fmap2.std().item(), cnet.mean().item(), cnet.std().item()], dtype=np.float64),
        pyr3=np32(corr_fn.corr_pyramid[3]), pyr2=np32(corr_fn.corr_pyramid[2]),
        pyr1_rows=np32(corr_fn.corr_pyramid[1][::37]), pyr0_rows=np32(corr_fn.corr_pyramid[0][::37]),
        look0=np32(look0[:1]), look_rand=np32(look_r[:1]), crand=np32(crand[:1]),
        look_rand_b1_sample=np32(look_r[-1:, ::9]),
        net1=np32(n1[:, ::8]), up_mask1=np32(up_mask1[:, ::16]), delta1=np32(delta1), up1=np32(up1),
        flow_low=np32(flow_low), flow_up=np32(flow_up),
        pred_first=np32(preds[0]), pred_mid=np32(preds[len(preds) // 2]),
        n_preds=np.array([len(preds)]),
    )
    if flow_init is not None:
        rec["flow_init"] = np32(flow_init)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, {k: v.shape for k, v in rec.items()}, "max|flow_up|", float(flow_up.abs().max()))


def case_train(net, name, inputs, iters):
    """Train-mode forward + sequence-L1 + backward (raft.py + losses.py:18-47)."""
    image1, image2, mask1, mask2 = inputs
    i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
    g = torch.Generator().manual_seed(5)
    flow_gt = (torch.randn(image1.shape[0], 2, *image1.shape[-2:], generator=g) * 5).clamp(-400, 400)
    valid = torch.ones(image1.shape[0], *image1.shape[-2:])
    net.train()
    net.zero_grad()
    preds = net(i1, i2, m1, m2, iters=iters)
    loss, _ = orc.sequence_l1(preds, flow_gt, valid)
    loss.backward()
    keys = ["fnet.conv1.weight", "fnet.mask_conv1.weight", "fnet.fusion1.mask2img.conv.weight",
            "fnet.layer2.0.downsample.0.weight", "fnet.fusion5.mask2img.conv.bias",
            "cnet.layer3.1.conv2.weight", "cnet.norm1.weight", "cnet.layer2.0.norm3.bias",
            "update_block.encoder.convc1.weight", "update_block.gru.convq2.weight",
            "update_block.gru.convz1.bias", "update_block.flow_head.conv2.weight",
            "update_block.mask.2.weight", "update_block.encoder.convf1.weight"]
    params = dict(net.named_parameters())
    rec = dict(loss=np.array([loss.item()], dtype=np.float64), pred_last=np32(preds[-1]),
               flow_gt_crc=np.array([crc(flow_gt)], dtype=np.int64))
    for k in keys:
        gk = params[k].grad
        rec["grad:" + k] = np32(gk.flatten()[:: max(1, gk.numel() // 512)])
        rec["gnorm:" + k] = np.array([gk.norm().item()], dtype=np.float64)
    total = torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters()))
    rec["grad_total_norm"] = np.array([total.item()], dtype=np.float64)
    sd = net.state_dict()
    for k in ["cnet.norm1.running_mean", "cnet.norm1.running_var", "cnet.layer2.0.downsample.1.running_var",
              "cnet.mask_norm1.running_mean"]:
        rec["buf:" + k] = np32(sd[k])
    assert all(p.grad is not None for p in net.parameters())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "loss", loss.item(), "gradnorm", total.item())
    net.zero_grad()


def sampler_taps(h, w, xs, y_fixed):
    """Which taps (and weights) does the reference sampler use along x?

    Runs the reference ``bilinear_sampler`` (utils.py:57-71) on an (N,1,h,w)
    plane for 9 window offsets per query, and reads the taps back from the
    gradient w.r.t. the plane (row-summed): lowest touched column + its weight.
    """
    n = xs.numel()
    r = 4
    off = torch.linspace(-r, r, 2 * r + 1)
    cx = xs.view(n, 1, 1) + off.view(1, 9, 1).expand(n, 9, 9)   # axis 0 = x-offset (corr.py:37-43)
    cy = torch.full((n, 9, 9), float(y_fixed))
    coords = torch.stack([cx, cy], -1)
    lo = np.full((n, 9), -9, dtype=np.int32)
    cnt = np.zeros((n, 9), dtype=np.int32)
    wlo = np.zeros((n, 9), dtype=np.float32)
    for a in range(9):
        plane = torch.zeros(n, 1, h, w, requires_grad=True)
        out = bilinear_sampler(plane, coords)
        out[:, 0, a, 4].sum().backward()
        gcol = plane.grad[:, 0].sum(1)  # (n, w)
        nz = gcol != 0
        cnt[:, a] = nz.sum(1).numpy()
        first = torch.where(nz.any(1), nz.float().argmax(1), torch.full((n,), -9))
        lo[:, a] = first.numpy()
        wlo[:, a] = torch.where(first >= 0, gcol[torch.arange(n), first.clamp(min=0)], torch.zeros(n)).numpy()
    return lo, cnt, wlo


def case_sampler_index():
    rec = {}
    g = torch.Generator().manual_seed(3)
    for (h, w) in [(48, 64), (24, 32), (12, 16), (6, 8), (46, 62), (23, 31), (11, 15), (5, 7), (68, 120), (16, 24)]:
        ints = torch.arange(0, w).float()
        xs = torch.cat([ints, ints / 2, ints / 4, ints / 8, ints + 0.5,
                        torch.rand(64, generator=g) * (w + 6) - 3])
        lo, cnt, wlo = sampler_taps(h, w, xs, y_fixed=h // 2)
        rec[f"xs_{h}x{w}"] = xs.numpy()
        rec[f"lo_{h}x{w}"] = lo
        rec[f"cnt_{h}x{w}"] = cnt
        rec[f"wlo_{h}x{w}"] = wlo
    # known-answer values: half-integer / out-of-range samples of a ramp plane
    plane = (torch.arange(48 * 64).float().view(1, 1, 48, 64) % 97) / 7.0
    pts = torch.tensor([[0.0, 0.0], [63.0, 47.0], [10.5, 3.25], [-0.5, 2.0], [63.5, 47.5], [-3.0, -3.0],
                        [70.0, 20.0], [5.0, 5.0], [31.999, 24.001]]).view(1, 9, 1, 2)
    rec["kat_plane"] = np32(plane)
    rec["kat_pts"] = np32(pts)
    rec["kat_out"] = np32(bilinear_sampler(plane, pts))
    np.savez_compressed(os.path.join(HERE, "sampler_index.npz"), **rec)
    print("sampler_index", len(rec))


def case_upsample(net):
    flow = torch.arange(2 * 2 * 5 * 7).float().view(2, 2, 5, 7) / 10 - 3
    g = torch.Generator().manual_seed(11)
    mask = torch.randn(2, 576, 5, 7, generator=g)
    np.savez_compressed(os.path.join(HERE, "upsample.npz"), flow=np32(flow), mask=np32(mask),
                        out=np32(net.upsample_flow(flow, mask)))


def main():
    net = build_reference()
    spec = [["flow_net." + k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_spec.json"), "w") as f:
        json.dump(spec, f)
    print("state_dict keys:", len(spec))

    case_forward(net, "fwd_rand_128x192_b2_it12", orc.synthetic_inputs(2, 128, 192, seed=0), iters=12)
    case_forward(net, "fwd_shift_128x192_b2_it12", orc.shifted_pair(2, 128, 192, seed=1), iters=12)
    g = torch.Generator().manual_seed(9)
    finit = torch.randn(1, 2, 16, 20, generator=g) * 2
    case_forward(net, "fwd_shift_128x160_b1_it4_init", orc.shifted_pair(1, 128, 160, seed=2), iters=4, flow_init=finit)
    case_train(net, "train_shift_128x128_b2_it3", orc.shifted_pair(2, 128, 128, seed=4), iters=3)
    net = build_reference()  # the train step updated the BN running statistics: start clean again
    case_sampler_index()
    case_upsample(net)

    # 384x512 B=1 (BASELINE config 1): checksums/statistics + a thin slice only
    net.eval()
    image1, image2, mask1, mask2 = orc.shifted_pair(1, 384, 512, seed=6)
    i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
    with torch.no_grad():
        flow_low, flow_up = net(i1, i2, m1, m2, iters=12, test_mode=True)
    np.savez_compressed(os.path.join(HERE, "fwd_shift_384x512_b1_it12.npz"),
                        in_crc=np.array([crc(image1), crc(image2), crc(mask1)], dtype=np.int64),
                        flow_low=np32(flow_low), flow_up_sub=np32(flow_up[:, :, ::4, ::4]),
                        flow_up_stats=np.array([flow_up.mean().item(), flow_up.abs().max().item()], dtype=np.float64))
    print("384x512 max|flow|", flow_up.abs().max().item())

    # concat fusion variant (17/36 reference configs): final flow only
    net_c = RAFT(in_channels=256, inside_fusion="parallel", fuse_cnet=True, cfg=cfg("concat"))
    sdc = net_c.state_dict()
    net_c.load_state_dict({k: det_tensor("flow_net." + k, v.shape) for k, v in sdc.items()})
    with open(os.path.join(HERE, "state_dict_spec_concat.json"), "w") as f:
        json.dump([["flow_net." + k, list(v.shape), str(v.dtype)] for k, v in sdc.items()], f)
    net_c.eval()
    image1, image2, mask1, mask2 = orc.shifted_pair(1, 128, 160, seed=8)
    i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
    with torch.no_grad():
        fl, fu = net_c(i1, i2, m1, m2, iters=4, test_mode=True)
    np.savez_compressed(os.path.join(HERE, "fwd_concat_128x160_b1_it4.npz"), flow_low=np32(fl), flow_up=np32(fu),
                        in_crc=np.array([crc(image1), crc(image2), crc(mask1)], dtype=np.int64))

    # SA / CA fusion variants (one reference config each): the fusion units alone on random (q, v), and the
    # final flow of the whole network at 128x160
    from FF_RAFT_Core.parallel_fusion import CA, SA  # noqa: E402  (reference)
    for ft, cls in (("SA", SA), ("CA", CA)):
        unit = cls(64)
        unit.load_state_dict({k: det_tensor(f"unit_{ft}." + k, v.shape) for k, v in unit.state_dict().items()})
        g = torch.Generator().manual_seed(21)
        q, v = torch.randn(2, 64, 12, 20, generator=g), torch.randn(2, 64, 12, 20, generator=g)
        with torch.no_grad():
            out = unit.eval()(q, v)
        net_v = RAFT(in_channels=256, inside_fusion="parallel", fuse_cnet=True, cfg=cfg(ft))
        sdv = net_v.state_dict()
        net_v.load_state_dict({k: det_tensor("flow_net." + k, v_.shape) for k, v_ in sdv.items()})
        with open(os.path.join(HERE, f"state_dict_spec_{ft.lower()}.json"), "w") as f:
            json.dump([["flow_net." + k, list(v_.shape), str(v_.dtype)] for k, v_ in sdv.items()], f)
        net_v.eval()
        image1, image2, mask1, mask2 = orc.shifted_pair(1, 128, 160, seed=8)
        i1, i2, m1, m2 = orc.prepare_inputs(image1, image2, mask1, mask2, 3)
        with torch.no_grad():
            fl, fu = net_v(i1, i2, m1, m2, iters=4, test_mode=True)
        np.savez_compressed(os.path.join(HERE, f"fwd_{ft.lower()}_128x160_b1_it4.npz"), flow_low=np32(fl), flow_up=np32(fu),
                            unit_out=np32(out), unit_keys=np.array(list(unit.state_dict().keys())),
                            in_crc=np.array([crc(image1), crc(image2), crc(mask1)], dtype=np.int64))
        print(ft, "unit |out|", out.abs().max().item(), "flow", fu.abs().max().item())


if __name__ == "__main__":
    main()
