"""max |flow_up - oracle| (px) on several 384x512 pairs, 12 iterations, for the routes of the update block:
split-pair activations on / off.  The oracle runs on the host cores (a few seconds per pair)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ffraft_ref as orc
from oracle.weights import det_tensor
import bench
from focusflow_official_amd import FF_RAFT_FUSION, update_block

sd = {k: det_tensor(k, s) for k, s, _ in json.load(open(os.path.join(bench.ROOT, "tests", "golden", "state_dict_spec.json")))}
m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=bench.cfg())
m.load_state_dict(sd, strict=True)
m = m.cuda().eval()
torch.set_num_threads(bench.host_cores())
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    inp = orc.shifted_pair(1, 384, 512, seed=100 + seed, shift=((seed * 3) % 7 - 3, (seed * 5) % 9 - 4))
    with torch.no_grad():
        ref = orc.ffraft_forward(sd, *inp, raft_iters=12, test_mode=True)[1]
        out = {}
        for on in (True, False):
            update_block._SPLIT_ACT = on
            fu = m(*[t.cuda() for t in inp], raft_iters=12, test_mode=True)[1].cpu()
            out[on] = (float((fu - ref).abs().max()), float(torch.sqrt(((fu - ref) ** 2).sum(1)).mean()))
    print(f"seed {seed}: |flow| max {float(ref.abs().max()):6.2f} px | split on: max {out[True][0]:.2e} epe {out[True][1]:.2e} | off: max {out[False][0]:.2e} epe {out[False][1]:.2e}", flush=True)
