"""Training across ranks: the module's parameters are ordinary nn.Parameters, so the
reference's DistributedDataParallel wrap (common.py:45-50, find_unused_parameters=False) works
unchanged.  Two ranks share cuda:0 here and talk over gloo (RCCL refuses two ranks on one device; the
driver's 8-GPU run uses backend "nccl" = RCCL).  Check: each rank runs half of a 2-pair batch,
`loss *= world_size` (train.py:313-314) + DDP's mean = the single-process full-batch gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_spec

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, partial):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from argparse import Namespace
    from torch.nn.parallel import DistributedDataParallel
    from focusflow_official_amd import FF_RAFT_FUSION
    from oracle import ffraft_ref as orc
    from oracle.weights import det_tensor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                    MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
    sd = {k: det_tensor(k, s) for k, s, _ in golden_spec()}
    model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    model.flow_net.freeze_bn()      # per-replica BatchNorm statistics would differ from the full batch
    ddp = DistributedDataParallel(model, device_ids=[0], output_device=0, find_unused_parameters=False)
    inp = orc.shifted_pair(2, 128, 128, seed=21)
    mine = [t[rank:rank + 1].to(dev) for t in inp]
    preds = ddp(*mine, raft_iters=2)
    # partial: the loss reaches only the last prediction, so some applications of the shared update-block convs never
    # run their backward - DDP (find_unused_parameters=False) must still see one gradient per parameter (fn.ParamGate)
    loss = preds[-1].abs().mean() if partial else sum(p.abs().mean() for p in preds)
    (loss * world).backward()                                   # train.py:313-316
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())
    if rank == 0:
        torch.save(grads, os.path.join(out_dir, "ddp_grads.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("partial", [False, True])
def test_ddp_two_ranks_match_full_batch(tmp_path, partial):
    from argparse import Namespace
    from focusflow_official_amd import FF_RAFT_FUSION
    from oracle import ffraft_ref as orc
    from oracle.weights import det_tensor
    mp.spawn(_worker, args=(2, 29533 + int(partial), str(tmp_path), partial), nprocs=2, join=True)
    ddp_grads = torch.load(os.path.join(tmp_path, "ddp_grads.pt"))
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                    MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))
    sd = {k: det_tensor(k, s) for k, s, _ in golden_spec()}
    model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    model.load_state_dict(sd)
    model = model.to("cuda:0").train()
    model.flow_net.freeze_bn()
    inp = [t.to("cuda:0") for t in orc.shifted_pair(2, 128, 128, seed=21)]
    preds = model(*inp, raft_iters=2)
    # full batch: mean over 2 samples of per-sample means == sum of the two ranks' losses / 2 * ... :
    # rank loss_r = sum_i mean_over_sample_r(|p_i|); DDP averages world*loss_r over ranks = sum_r loss_r
    loss = sum(torch.stack([p[r].abs().mean() for r in range(2)]).sum() for p in (preds[-1:] if partial else preds))
    loss.backward()
    report = []
    for k, p in model.named_parameters():
        ref = p.grad.cpu()
        scale = ref.abs().max().item()
        # biases that feed an InstanceNorm have an exactly-zero true gradient (only rounding noise): skip
        if scale < 1e-5:
            continue
        report.append((((ddp_grads[k] - ref).abs().max() / scale).item(), k, scale))
    report.sort(reverse=True)
    # both sides are the HIP path: what differs is the order of the atomic weight-gradient adds and the per-rank batch
    # of 1 instead of 2 (other tiles, other split scales of the gradients)
    print(report[:4])
    assert report[0][0] < 1e-4, report[:8]      # measured: 1e-5 (it was asserted at 1e-2)
