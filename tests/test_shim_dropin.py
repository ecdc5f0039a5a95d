"""The drop-in boundary as the reference's scripts use it: with the repo root on sys.path,
`from FF_RAFT_Core.ff_raft import FF_RAFT_FUSION` (core/models/ff-raft/train.py:19) and
`from PWCNet_Core.ff_pwcnet import FF_PWCNET` (core/models/ff-pwcnet/train.py:19) must hand back the MI355X modules,
and the call sequence of train.py must run on them: constructor with train.py:186-188's keyword arguments,
`model.flow_net.freeze_bn()` (:193), `.to(device)`, AdamW over the trainable parameters (:211-214), the DDP wrap of
common.py:45-50 (find_unused_parameters=False) over a real RCCL process group, and one step of the loop at :291-328
(forward, loss, `loss *= world_size`, GradScaler-scaled backward, clip, step)."""
import os
import sys
from argparse import Namespace

import pytest
import torch

from conftest import ROOT, golden_spec

DEV = "cuda:0"


def test_shim_packages_resolve_to_the_hip_modules():
    """CPU: the import paths of both train.py scripts resolve, and to THIS package's classes."""
    assert ROOT in sys.path
    from FF_RAFT_Core.ff_raft import FF_RAFT_FUSION
    from FF_RAFT_Core.corr import CorrBlock
    from FF_RAFT_Core.raft import RAFT
    from FF_RAFT_Core.update import BasicUpdateBlock
    from FF_RAFT_Core.parallel_fusion import BasicParallelFusionLayer
    from PWCNet_Core.ff_pwcnet import FF_PWCNET
    import focusflow_official_amd as pkg
    assert FF_RAFT_FUSION is pkg.FF_RAFT_FUSION and CorrBlock is pkg.CorrBlock
    assert RAFT.__module__ == "focusflow_official_amd.raft_net"
    assert BasicUpdateBlock.__module__ == "focusflow_official_amd.update_block"
    assert BasicParallelFusionLayer.__module__ == "focusflow_official_amd.cce"
    assert FF_PWCNET.__module__ == "focusflow_official_amd.pwcnet"


@pytest.fixture(scope="module")
def rccl_group():
    """world_size 1 over backend "nccl" (= RCCL on ROCm): loads librccl, creates a communicator and all-reduces."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.full((1024,), 3.0, device=DEV)
    dist.all_reduce(t)
    assert float(t.sum()) == 3.0 * 1024
    yield dist
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ffraft_train_py_call_sequence_through_the_shim(rccl_group):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from FF_RAFT_Core.ff_raft import FF_RAFT_FUSION                     # train.py:19
    from focusflow_official_amd.losses import build_losses              # train.py:23 `from losses import build_losses`
    from oracle import ffraft_ref as orc
    from oracle.weights import det_tensor
    world_size, rank, local_rank = 1, 0, 0
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point", STAGE="things", CLIP=1.0),
                    MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False, PRETRAIN=None, FUSION="parallel",
                                    FUSION_CHANNEL=256, SMALL=False, DROPOUT=0.0, ALT_CORR=False, ABANDON_FNET=False,
                                    FUSE_CNET=True, FREEZE_MODULE=False, ITERS=3))
    model = FF_RAFT_FUSION(pretrain=cfg.MODEL.PRETRAIN, load_raft=None, use_fusion=cfg.MODEL.FUSION,      # train.py:186-188
                           fusion_channels=cfg.MODEL.FUSION_CHANNEL, raft_small=cfg.MODEL.SMALL, dropout=cfg.MODEL.DROPOUT,
                           alternate_corr=cfg.MODEL.ALT_CORR, abandon_fnet=cfg.MODEL.ABANDON_FNET, fuse_cnet=cfg.MODEL.FUSE_CNET,
                           freeze_flownet=cfg.MODEL.FREEZE_MODULE, cfg=cfg)
    if cfg.TRAIN.STAGE != "chairs":
        model.flow_net.freeze_bn()                                      # train.py:192-193
    model.load_state_dict({k: det_tensor(k, s) for k, s, _ in golden_spec()}, strict=True)   # train.py:199 (checkpoint)
    assert hasattr(model, "fusion_layer") or hasattr(model, "flow_net")  # train.py:221,225 touch these
    device = torch.device("cuda", local_rank)
    model.to(device)
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=4e-4, weight_decay=1e-5, eps=1e-8)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, 4e-4, 100, pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
    model = DDP(model, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=False)    # common.py:45-50
    loss_function = build_losses("MixLoss", gamma=0.8, max_flow=400, kernel_size=1, sigma=0.01, lamda=1)
    scaler = torch.cuda.amp.GradScaler(enabled=False)
    image1, image2, mask1, mask2 = [x.cuda() for x in orc.shifted_pair(2, 128, 160, seed=5)]
    flow = torch.randn(2, 2, 128, 160, generator=torch.Generator().manual_seed(1)).cuda() * 3
    valid = torch.ones(2, 128, 160).cuda()
    before = {k: v.detach().clone() for k, v in model.module.named_parameters()}
    model.train()
    optimizer.zero_grad()
    with torch.cuda.amp.autocast(enabled=False):                        # MIXED_PRECISION: false in every shipped config
        flow_predictions = model(image1, image2, mask1, mask2, raft_iters=cfg.MODEL.ITERS)
        loss, metrics = loss_function(flow_predictions, flow, valid, mask1)
    assert isinstance(flow_predictions, list) and len(flow_predictions) == 3 and flow_predictions[0].shape == (2, 2, 128, 160)
    if rank != -1:
        loss *= world_size
    scaler.scale(loss).backward()
    scaler.unscale_(optimizer)
    torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.TRAIN.CLIP)
    scaler.step(optimizer)
    scheduler.step()
    scaler.update()
    assert torch.isfinite(loss) and "epe" in metrics
    moved = [k for k, v in model.module.named_parameters() if v.requires_grad and not torch.equal(v, before[k])]
    assert len(moved) > 200, "one optimiser step must move (almost) every trainable parameter"
    # evaluate.py's call: eval mode, test_mode=True -> (flow_low, flow_up)
    model.eval()
    with torch.no_grad():
        flow_low, flow_up = model.module(image1, image2, mask1, mask2, raft_iters=2, test_mode=True)
    assert flow_low.shape == (2, 2, 16, 20) and flow_up.shape == (2, 2, 128, 160)


@pytest.mark.gpu
def test_ffpwcnet_train_py_call_sequence_through_the_shim(rccl_group):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from PWCNet_Core.ff_pwcnet import FF_PWCNET                          # ff-pwcnet/train.py:19
    from focusflow_official_amd.pwc_losses import build_losses
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point", STAGE="chairs", CLIP=1.0, LOSS_TYPE="MixLoss",
                                    LOSS_MODE="pretrain", LOSS_WEIGHTS=[0.005, 0.01, 0.02, 0.08, 0.32], LOSS_Q=0.4,
                                    LOSS_EPSILON=0.01, LOSS_KERNEL_SIZE=5, LOSS_SIGMA=1.7, LOSS_LAMDA=0.7),
                    MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
    torch.manual_seed(0)
    model = FF_PWCNET(cfg, pretrain=None, load_pwcnet=None)              # ff-pwcnet/train.py:184
    model.to(torch.device("cuda", 0))
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, weight_decay=4e-4, eps=1e-8)
    model = DDP(model, device_ids=[0], output_device=0, find_unused_parameters=False)
    loss_function = build_losses(cfg)
    g = torch.Generator().manual_seed(6)
    base = torch.rand(2, 3, 36, 52, generator=g)
    image1 = (torch.nn.functional.interpolate(base, size=(128, 192), mode="bilinear", align_corners=False) * 255).cuda()
    image2 = torch.roll(image1, shifts=(2, -3), dims=(2, 3))
    mask1 = ((torch.rand(2, 1, 128, 192, generator=g) < 0.02).float() * 255).cuda()
    mask2 = torch.zeros_like(mask1)
    flow = (torch.randn(2, 2, 128, 192, generator=g) * 3).cuda()
    sparse = False
    model.train()
    optimizer.zero_grad()
    flow_predictions = model(image1, image2, mask1, mask2)              # ff-pwcnet/train.py:310
    loss, metrics = loss_function(flow_predictions, flow, mask1, sparse)
    loss *= 1
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.TRAIN.CLIP)
    optimizer.step()
    assert len(flow_predictions) == 5 and torch.isfinite(loss) and "epe" in metrics
    assert all(p.grad is not None for p in model.parameters())
    model.eval()
    with torch.no_grad():
        out = model.module(image1, image2, mask1, mask2, test_mode=True)
    assert out.shape == (2, 2, 128, 192)
