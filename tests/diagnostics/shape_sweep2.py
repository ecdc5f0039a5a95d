"""Second robustness sweep: mask modes at odd sizes, batch extremes, iteration counts, flow_init, non-contiguous inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
from focusflow_official_amd import FF_RAFT_FUSION
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor

torch.set_num_threads(16)


def build(modal="point", ft="1x1conv"):
    cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL=modal, MASK_DILATE=31, KERNEL_SIZE=31, KERNEL_SIGMA=5.0),
                    MODEL=Namespace(FUSION_TYPE=ft, LOAD_MODULE_TO_BRANCH=False))
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg)
    sd = {k: det_tensor(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    return m.cuda().eval(), sd, cfg


def check(tag, m, sd, inp, iters=3, flow_init=None, **kw):
    with torch.no_grad():
        fl, fu = m(*[t.cuda() for t in inp], raft_iters=iters, flow_init=None if flow_init is None else flow_init.cuda(), test_mode=True)
        rl, ru = orc.ffraft_forward(sd, *inp, raft_iters=iters, flow_init=flow_init, test_mode=True, **kw)
    d = (fu.cpu() - ru).abs().max().item()
    print(f"{tag}: {d:.2e} {'OK' if d < 1e-3 else 'FAIL'}", flush=True)


m, sd, _ = build()
check("B16 128x192", m, sd, orc.shifted_pair(16, 128, 192, seed=1))
check("B1 384x512 it1", m, sd, orc.shifted_pair(1, 384, 512, seed=2), iters=1)
check("B1 128x160 it24", m, sd, orc.shifted_pair(1, 128, 160, seed=3), iters=24)
g = torch.Generator().manual_seed(5)
check("flow_init 136x152", m, sd, orc.shifted_pair(2, 136, 152, seed=4), flow_init=torch.randn(2, 2, 17, 19, generator=g) * 2)
# non-contiguous inputs: channels-last memory format and a spatial slice of a larger tensor
i1, i2, m1, m2 = orc.shifted_pair(1, 144, 208, seed=6)
big = torch.zeros(1, 3, 160, 224); big[:, :, 8:152, 8:216] = i1
nc = [big[:, :, 8:152, 8:216], i2.contiguous(memory_format=torch.channels_last), m1, m2]
with torch.no_grad():
    fu_nc = m(*[t.cuda() for t in nc], raft_iters=2, test_mode=True)[1]
    fu_c = m(*[t.cuda() for t in (i1, i2, m1, m2)], raft_iters=2, test_mode=True)[1]
print("non-contiguous inputs identical:", torch.equal(fu_nc, fu_c), flush=True)
for modal in ("frame", "neighborG", "neighborE", "context"):
    try:
        mm, sdd, cfg = build(modal)
        inp = orc.shifted_pair(1, 136, 152, seed=7)
        with torch.no_grad():
            fu = mm(*[t.cuda() for t in inp], raft_iters=2, test_mode=True)[1]
            ref = orc.ffraft_forward(sdd, *inp, raft_iters=2, test_mode=True, mask_modal=modal, cfg=cfg)[1] if "mask_modal" in orc.ffraft_forward.__code__.co_varnames else None
        print(f"mask mode {modal} 136x152: finite {bool(torch.isfinite(fu).all())}" + ("" if ref is None else f" diff {(fu.cpu() - ref).abs().max().item():.2e}"), flush=True)
    except Exception as e:
        print(f"mask mode {modal}: EXC {type(e).__name__}: {str(e)[:150]}", flush=True)
