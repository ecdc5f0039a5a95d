"""Phase times of the blocks of conv_dma.hip's fp32-input route on the encoders' layer shapes (a library whose conv_dma.o was
built with -DFF_DMA_STAMPS: FF_LAB_LIB=libfocusflow_stamps.so).  Stamps: block start | behind the first chunk's barrier +
conversion | end of the main loop | behind the last store."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from focusflow_official_amd import _hip, ops
DEV = "cuda:0"
ws = torch.zeros(1 << 21, dtype=torch.int64, device=DEV)
orig = _hip.call
def call(name, *args):
    if name == "ff_conv2d_fwd":
        args[0]._obj.splitk_ws = ws.data_ptr()
    return orig(name, *args)
_hip.call = call
ops._hip.call = call
g = torch.Generator().manual_seed(0)
for name, b, h, w, cin, cout, inorm in [("64->64 192x256 x16", 16, 192, 256, 64, 64, False), ("64->64 inorm", 16, 192, 256, 64, 64, True), ("96->96 96x128 x16", 16, 96, 128, 96, 96, False),
                                        ("128->128 48x64 x16", 16, 48, 64, 128, 128, False), ("256->192 46x62 x8", 8, 46, 62, 256, 192, False)]:
    x = torch.randn(b, h, w, cin, generator=g).to(DEV)
    wp = torch.empty(cout, 9 * cin, device=DEV)
    ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(DEV), wp, cin)
    wp = ops.pack_split(wp)
    wf = ops.pack_frag16(wp, cout)
    kw = {}
    if inorm:
        kw = dict(in_scale=torch.rand(b, cin, generator=g).to(DEV) + 0.5, in_shift=torch.randn(b, cin, generator=g).to(DEV), in_act=1)
    for _ in range(3):
        ops.conv2d([x], wp, None, cout, 3, 3, 1, (1, 1), w_fmt=1, w_frag=wf, **kw)
    torch.cuda.synchronize()
    ws.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d([x], wp, None, cout, 3, 3, 1, (1, 1), w_fmt=1, w_frag=wf, **kw)
    e1.record()
    torch.cuda.synchronize()
    st = ws.cpu().numpy().reshape(-1, 4)
    st = st[st[:, 0] > 0].astype(np.float64) / 100.0
    t0 = st[:, 0].min()
    pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    print(f"{name:20s} blocks {len(st):5d} launch {e0.elapsed_time(e1) * 1e3:6.1f} us | prologue {np.median(pro):5.2f} (max {pro.max():5.1f}) | loop {np.median(loop):5.2f} "
          f"(min {loop.min():5.2f} max {loop.max():5.1f}) | epilogue {np.median(epi):5.2f} (max {epi.max():5.1f}) | block total {np.median(st[:, 3] - st[:, 0]):5.2f} | first start -> last end {st[:, 3].max() - t0:6.1f} us")
