"""Phase times of gru_pass.hip's blocks (LAB build: tools/build_lab.sh, FF_LAB_LIB=libfocusflow_lab.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from focusflow_official_amd import _hip, ops

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W, c = 48, 64, 128
lib = _hip.load()
ws = torch.zeros(1 << 16, dtype=torch.int64, device=DEV)
lib.ff_lab_gru_pass_stamps.argtypes = [ctypes.c_void_p]
lib.ff_lab_gru_pass_stamps(ctypes.c_void_p(ws.data_ptr()))
g = torch.Generator().manual_seed(0)
h = torch.randn(B, H, W, c, generator=g).to(DEV)
mo = torch.randn(B, H, W, c, generator=g).to(DEV)
hs, ms = ops.split_copy(h), ops.split_copy(mo)
przr, prq = torch.randn(B, H, W, 2 * c, generator=g).to(DEV), torch.randn(B, H, W, c, generator=g).to(DEV)
for d, (kh, kw) in enumerate(((1, 5), (5, 1))):
    def pack(co):
        wp = torch.empty(co, kh * kw * 2 * c, device=DEV)
        ops.pack_conv_weight((torch.randn(co, 2 * c, kh, kw, generator=g) / 36).to(DEV), wp, 2 * c)
        return ops.pack_frag16(ops.pack_split(wp), co)
    fzr, fq = pack(2 * c), pack(c)
    bzr, bq = torch.randn(2 * c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    for _ in range(5):
        ops.gru_pass(d, hs, ms, h, przr, prq, fzr, fq, bzr, bq, 1)
    torch.cuda.synchronize()
    ws.zero_()
    ops.gru_pass(d, hs, ms, h, przr, prq, fzr, fq, bzr, bq, 1)
    torch.cuda.synchronize()
    st = ws.cpu().numpy()[:65535].reshape(-1, 5)
    st = st[st[:, 0] > 0].astype(np.float64) / 100.0
    t0 = st[:, 0].min()
    d1, d2, d3, d4 = (st[:, i + 1] - st[:, i] for i in range(4))
    print(f"pass {d + 1}: blocks {len(st)}: start spread {st[:, 0].max() - t0:4.1f} us | z|r loop {np.median(d1):5.1f} (max {d1.max():5.1f}) | r*h -> LDS {np.median(d2):4.1f} (max {d2.max():4.1f}) | "
          f"q loop {np.median(d3):5.1f} (max {d3.max():5.1f}) | epilogue {np.median(d4):4.1f} (max {d4.max():4.1f}) | first start -> last end {st[:, 4].max() - t0:5.1f} us")
