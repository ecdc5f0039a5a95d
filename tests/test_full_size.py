"""BASELINE configs[2] and configs[3] at their FULL sizes on the GPU (the per-op and small-shape tests live in
test_hip_parity / test_hip_backward / test_hip_pwc):
  * one FF-RAFT training step at 8 x 368x496, 12 iterations, MixLoss (ffraft_chairs_orb.yaml:35-39) - the per-GPU share
    of the 8-GPU DDP configuration;
  * FF_PWCNET at 1 x 448x1024 (KITTI-shaped, SIFT-density mask) against oracle/pwc_ref.py."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ffraft_ref as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_config3_training_step_8x368x496_it12(det_sd):
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd.losses import build_losses
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                    MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    m.load_state_dict(det_sd, strict=True)
    m = m.to(DEV).train()
    crit = build_losses("MixLoss", gamma=0.8, max_flow=400, kernel_size=1, sigma=0.01, lamda=1)
    b, h, w = 8, 368, 496
    inp = [t.to(DEV) for t in orc.shifted_pair(b, h, w, seed=31)]
    g = torch.Generator().manual_seed(32)
    flow_gt = (torch.randn(b, 2, h, w, generator=g) * 5).clamp(-400, 400).to(DEV)
    valid = torch.ones(b, h, w, device=DEV)
    with torch.no_grad():                      # the inference kernels (fused epilogues, normalise-on-load, GRU context hoist)
        preds0 = m(*inp, raft_iters=12)
        loss0, _ = crit(preds0, flow_gt, valid, inp[2])
    preds = m(*inp, raft_iters=12)             # the recorded path (one autograd node per launch group)
    loss, metrics = crit(preds, flow_gt, valid, inp[2])
    assert len(preds) == 12 and preds[0].shape == (b, 2, h, w)
    assert torch.isfinite(loss) and abs(loss.item() - loss0.item()) <= 2e-5 * abs(loss0.item()), (loss.item(), loss0.item())
    err = max(float((a - c).abs().max()) for a, c in zip(preds, preds0))
    assert err < 1e-3, f"recorded vs inference forward differ by {err:.3e} px"
    loss.backward()
    missing = [k for k, p in m.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:5]
    bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    assert not bad, bad[:5]
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    assert torch.isfinite(gn) and gn > 0


def test_config3_gradients_8x368x496_it12_against_cpu_autograd(det_sd):
    """The per-GPU share of BASELINE configs[2] - EIGHT pairs of 368 x 496, 12 iterations - through forward and backward, the
    gradients of sixteen named parameters against CPU autograd through the oracle (fp32, one run: ~1 min on the box's host
    cores).  The one-pair test below cannot see a cross-sample slip of a backward kernel (a batch stride, a per-image table
    indexed by the wrong image, the T x B stacking of the recorded update loop); here every sample contributes a different
    gradient and a slip shows up as O(1) of the tensor's maximum.  Bound: 2e-2 of the maximum (the whole-network rule of
    test_hip_backward._check_grad_spread - the test weights make the recurrence ill-conditioned, see there)."""
    from focusflow_official_amd import FF_RAFT_FUSION
    from test_hip_backward import _cfg, _oracle_grads
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg())
    m.load_state_dict(det_sd, strict=True)
    m = m.to(DEV).train()
    m.flow_net.freeze_bn()
    b, h, w, iters = 8, 368, 496, 12
    inp = orc.shifted_pair(b, h, w, seed=61)
    gen = torch.Generator().manual_seed(62)
    flow_gt = (torch.randn(b, 2, h, w, generator=gen) * 5).clamp(-400, 400)
    valid = torch.ones(b, h, w)
    loss_fn = lambda preds: orc.sequence_l1(preds, flow_gt.to(preds[0]), valid.to(preds[0]))[0]  # noqa: E731
    preds = m(*[t.to(DEV) for t in inp], raft_iters=iters)
    loss = loss_fn(preds)
    loss.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    g32, last32 = _oracle_grads(det_sd, inp, iters, loss_fn, torch.float32)
    err = float((preds[-1].detach().cpu() - last32).abs().max())
    assert err <= 2e-3, f"last prediction: {err:.3e} px from the oracle's fp32 run"
    params = dict(m.named_parameters(remove_duplicate=False))
    names = ["flow_net.fnet.conv1.weight", "flow_net.fnet.layer1.0.conv1.weight", "flow_net.fnet.layer2.0.conv2.weight", "flow_net.fnet.conv2.weight",
             "flow_net.fnet.fusion3.img2mask.conv.weight", "flow_net.cnet.layer2.0.downsample.0.weight", "flow_net.cnet.norm1.weight",
             "flow_net.cnet.mask_layer3.0.conv1.weight", "flow_net.update_block.encoder.convc1.weight", "flow_net.update_block.encoder.convf1.weight",
             "flow_net.update_block.encoder.conv.weight", "flow_net.update_block.gru.convq1.weight", "flow_net.update_block.gru.convz2.weight",
             "flow_net.update_block.flow_head.conv1.weight", "flow_net.update_block.flow_head.conv2.weight", "flow_net.update_block.mask.2.weight"]
    report = []
    for name in names:
        got, want = params[name].grad.cpu().double(), g32[name].double()
        scale = float(want.abs().max())
        rel = float((got - want).abs().max()) / scale
        report.append(f"{name}: {rel:.2e} of max")
    print("\n".join(report))
    for name, line in zip(names, report):
        assert float(line.split(": ")[1].split(" ")[0]) <= 2e-2, "\n".join(report)


def test_full_resolution_backward_1x368x496_it12_against_cpu_autograd(det_sd):
    """The backward at FULL resolution (one FlyingChairs-sized pair, 12 iterations, frozen BatchNorm as train.py:192-193
    runs the later stages): loss, last prediction and sampled parameter gradients against CPU autograd through the oracle
    in fp32 AND fp64.  The 46 x 62 planes of 1/8 resolution (odd multiples of nothing: 23 x 31, 11 x 15, 5 x 7 below) and
    the ragged last tiles of every tiled backward kernel only exist at this size; the small-shape gradient tests cannot
    see an indexing slip there.  Bound per tensor: 8 x the oracle's own fp32-vs-fp64 spread, at least 5e-3 of the tensor's
    maximum (the rule of test_hip_backward._check_grad_spread, where it is derived)."""
    from focusflow_official_amd import FF_RAFT_FUSION
    from test_hip_backward import _cfg, _check_grad_spread, _oracle_grads
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg())
    m.load_state_dict(det_sd, strict=True)
    m = m.to(DEV).train()
    m.flow_net.freeze_bn()
    h, w, iters = 368, 496, 12
    inp = orc.shifted_pair(1, h, w, seed=51)
    gen = torch.Generator().manual_seed(52)
    flow_gt = (torch.randn(1, 2, h, w, generator=gen) * 5).clamp(-400, 400)
    valid = torch.ones(1, h, w)
    loss_fn = lambda preds: orc.sequence_l1(preds, flow_gt.to(preds[0]), valid.to(preds[0]))[0]  # noqa: E731
    preds = m(*[t.to(DEV) for t in inp], raft_iters=iters)
    loss = loss_fn(preds)
    loss.backward()
    torch.cuda.synchronize()
    g32, last32 = _oracle_grads(det_sd, inp, iters, loss_fn, torch.float32)
    g64, last64 = _oracle_grads(det_sd, inp, iters, loss_fn, torch.float64)
    with torch.no_grad():
        ref_loss = float(loss_fn([p.float() for p in orc.ffraft_forward(det_sd, *inp, raft_iters=iters, training=False)]))
    assert abs(loss.item() - ref_loss) <= 2e-4 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    err = float((preds[-1].detach().cpu() - last32).abs().max())
    spread = float((last32.double() - last64).abs().max())
    assert err <= max(1e-3, 4 * spread), f"last prediction: {err:.3e} px from the oracle (its own fp32-vs-fp64 spread {spread:.3e})"
    params = dict(m.named_parameters(remove_duplicate=False))
    assert not [k for k, p in params.items() if p.requires_grad and p.grad is None]
    names = ["flow_net.fnet.conv1.weight", "flow_net.fnet.mask_conv1.weight", "flow_net.fnet.layer1.0.conv1.weight",
             "flow_net.fnet.layer2.0.conv2.weight", "flow_net.fnet.layer3.1.conv2.weight", "flow_net.fnet.conv2.weight",
             "flow_net.fnet.fusion3.img2mask.conv.weight", "flow_net.cnet.layer2.0.downsample.0.weight", "flow_net.cnet.norm1.weight",
             "flow_net.cnet.mask_layer3.0.conv1.weight", "flow_net.update_block.encoder.convc1.weight", "flow_net.update_block.encoder.convf1.weight",
             "flow_net.update_block.gru.convq1.weight", "flow_net.update_block.gru.convz2.weight", "flow_net.update_block.flow_head.conv2.weight",
             "flow_net.update_block.mask.2.weight", "flow_net.update_block.mask.0.bias"]
    for name in names:
        _check_grad_spread(params[name].grad.cpu(), g32[name], g64[name], name)


def test_config4_ffpwcnet_1x448x1024_vs_oracle():
    from focusflow_official_amd.pwcnet import FF_PWCNET
    from oracle import pwc_ref
    from test_hip_pwc import _pwc_weights
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
    sd = _pwc_weights()
    m = FF_PWCNET(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(41)
    h, w = 448, 1024
    base = torch.rand(1, 3, h // 8 + 2, w // 8 + 2, generator=g)
    i1 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False) * 255
    i2 = torch.roll(i1, shifts=(3, -6), dims=(2, 3))
    m1 = (torch.rand(1, 1, h, w, generator=g) < 2000.0 / (h * w)).float() * 255      # SIFT-like density
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = pwc_ref.ffpwc_forward(sd, i1, i2, m1)
        ref_full = pwc_ref.ffpwc_forward(sd, i1, i2, m1, test_mode=True)
        got = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV))
        got_full = m(i1.to(DEV), i2.to(DEV), m1.to(DEV), torch.zeros_like(m1).to(DEV), test_mode=True)
    assert [tuple(t.shape) for t in got] == [(1, 2, 112, 256), (1, 2, 56, 128), (1, 2, 28, 64), (1, 2, 14, 32), (1, 2, 7, 16)]
    assert float(ref_full.abs().max()) > 0.05, "degenerate test: flow is ~0"
    for lvl, (a, r) in enumerate(zip(got, ref)):
        err = float((a.cpu() - r).abs().max())
        assert err <= 2e-4 * max(1.0, float(r.abs().max())), f"flow level {lvl + 2}: max err {err:.3e}"
    err = float((got_full.cpu() - ref_full).abs().max())
    assert got_full.shape == (1, 2, h, w) and err <= 2e-4 * max(1.0, float(ref_full.abs().max())), err
