"""Phase times of conv_dma.hip's blocks (lab build only: FF_HIPCC_EXTRA_conv_dma=-DFF_DMA_STAMPS python -m focusflow_official_amd.build).
Every block stamps s_memrealtime (100 MHz) at its start, behind its first barrier (prologue: first patch + weights landed),
at the end of its main loop and behind its last store; this tool prints the medians and the launch-wide spread.
    python tools/dma_stamps.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from focusflow_official_amd import _hip, ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 48, 64
ws = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
orig = _hip.call


def call(name, *args):
    if name == "ff_conv2d_fwd":
        args[0]._obj.splitk_ws = ws.data_ptr()
    return orig(name, *args)


_hip.call = call
ops._hip.call = call
g = torch.Generator().manual_seed(0)
for name, segs, cout, kh, kw, ep in [("zr 1x5", [128, 128], 256, 1, 5, "rh"), ("q 1x5", [128, 128], 128, 1, 5, "blend"), ("convc2 3x3", [256], 192, 3, 3, None),
                                     ("heads 3x3", [128], 512, 3, 3, None)]:
    cin = sum(segs)
    sp = [ops.split_copy(torch.randn(B, H, W, c, generator=g).to(DEV)) for c in segs]
    wp = torch.empty(cout, kh * kw * cin, device=DEV)
    ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5).to(DEV), wp, cin)
    wp = ops.pack_split(wp)
    wf = ops.pack_frag16(wp, cout) if os.environ.get('FRAG', '1') != '0' else None
    bias = torch.randn(cout, generator=g).to(DEV)
    res = torch.randn(B, H, W, cout, generator=g).to(DEV)
    hprev, z = torch.randn(B, H, W, 128, generator=g).to(DEV), torch.rand(B, H, W, 128, generator=g).to(DEV)
    kwargs = dict(res=res, act_res=2, ep_rh=hprev, ep_split=128, y_split=128) if ep == "rh" else \
        dict(res=res, act_res=3, ep_blend=(z, hprev), y2_split=True) if ep == "blend" else dict(act=1, y_split=cout % 32 == 0)
    for _ in range(5):
        ops.conv2d(sp, wp, bias, cout, kh, kw, 1, (kh // 2, kw // 2), w_fmt=1, w_frag=wf, **kwargs)
    torch.cuda.synchronize()
    ws.zero_()
    ops.conv2d(sp, wp, bias, cout, kh, kw, 1, (kh // 2, kw // 2), w_fmt=1, w_frag=wf, **kwargs)
    torch.cuda.synchronize()
    st = ws.cpu().numpy().reshape(-1, 4)
    st = st[st[:, 0] > 0].astype(np.float64) / 100.0          # microseconds
    t0 = st[:, 0].min()
    pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    print(f"{name:12s} blocks {len(st):5d}: start spread {st[:, 0].max() - t0:5.1f} us | prologue {np.median(pro):5.1f} (max {pro.max():5.1f}) | "
          f"loop {np.median(loop):5.1f} (min {loop.min():5.1f} max {loop.max():5.1f}) | epilogue {np.median(epi):5.1f} (max {epi.max():5.1f}) | "
          f"first start -> last end {st[:, 3].max() - t0:5.1f} us; median block end {np.median(st[:, 3]) - t0:5.1f}")
