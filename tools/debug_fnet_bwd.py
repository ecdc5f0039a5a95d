import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from focusflow_official_amd import FF_RAFT_FUSION, ops
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor
DEV='cuda:0'
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL='point'), MODEL=Namespace(FUSION_TYPE='1x1conv', LOAD_MODULE_TO_BRANCH=False))
spec = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests/golden/state_dict_spec.json')))
sd = {k: det_tensor(k, s) for k, s, _ in spec}
m = FF_RAFT_FUSION(use_fusion='parallel', fusion_channels=256, fuse_cnet=True, cfg=cfg); m.load_state_dict(sd); m = m.to(DEV).train()
inp = orc.shifted_pair(2, 128, 128, seed=4)
i1, i2, m1, m2 = orc.prepare_inputs(*inp, 3)
g = torch.Generator().manual_seed(0)
G = torch.randn(2, 256, 16, 16, generator=g)
for which, (im, mk) in (("call1 (image1, mask1)", (inp[0], inp[2])), ("call2 (image2, const mask)", (inp[1], None))):
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in sd.items()}
    ref = orc.cce_encoder(sdr, 'flow_net.fnet', i1 if mk is not None else i2, m1 if mk is not None else m2, 'instance', True)
    (ref * G).sum().backward()
    m.zero_grad()
    b, _, h, w = im.shape
    x = ops.prep_input(im.to(DEV), b, h, w, im.to(DEV))
    mm = ops.prep_input(mk.to(DEV), b, h, w, x) if mk is not None else ops.prep_input(None, b, h, w, x, fill=255.0)
    out = m.flow_net.fnet(x, mm)
    print(which, 'fwd err', (out.detach().cpu().permute(0,3,1,2) - ref.detach()).abs().max().item())
    (out * G.permute(0,2,3,1).contiguous().to(DEV)).sum().backward()
    params = dict(m.flow_net.fnet.named_parameters())
    worst = []
    for k, p in params.items():
        r = sdr['flow_net.fnet.' + k].grad
        if r is None or k.endswith('bias'): continue
        rel = ((p.grad.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-12)).item()
        worst.append((rel, k))
    worst.sort(reverse=True)
    print('  worst:', [(f'{a:.1e}', k) for a, k in worst[:10]])
    print('  best :', [(f'{a:.1e}', k) for a, k in worst[-4:]])
