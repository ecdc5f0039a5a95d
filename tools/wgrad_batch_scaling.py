import sys, os, torch
sys.path.insert(0, os.getcwd())
from focusflow_official_amd import ops
def run(b, h, w, cins, cout, kh, kw, pad):
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(b, h, w, c, generator=g).cuda() for c in cins]
    dy = (torch.randn(b, h, w, cout, generator=g) * 1e-4).cuda()
    gg, amax = ops.act_bwd(dy, None, 0, 1.0, cout, want_amax=True)
    cin = sum(cins)
    dw = torch.zeros(cout, kh * kw * cin, device="cuda"); db = torch.zeros(cout, device="cuda")
    for _ in range(3): ops.conv2d_wgrad(xs, gg, cout, kh, kw, 1, pad, g_amax=amax, want_db=True, dw=dw, db=db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): ops.conv2d_wgrad(xs, gg, cout, kh, kw, 1, pad, g_amax=amax, want_db=True, dw=dw, db=db)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, cins, cout, kh, kw, pad in [("zr 1x5", [128, 128], 256, 1, 5, (0, 2)), ("q 5x1", [128, 128], 128, 5, 1, (2, 0)), ("convc2 3x3", [256], 192, 3, 3, (1, 1)),
                                      ("heads 3x3", [128], 512, 3, 3, (1, 1)), ("conv 3x3", [192, 64], 126, 3, 3, (1, 1)), ("convc1 1x1", [352], 256, 1, 1, (0, 0)), ("mask2 1x1", [256], 576, 1, 1, (0, 0))]:
    t1 = run(8, 46, 62, cins, cout, kh, kw, pad)
    t12 = run(96, 46, 62, cins, cout, kh, kw, pad)
    print(f"{name:12s} B=8: {t1:7.1f} us  x12 = {12*t1:8.1f} us   B=96: {t12:8.1f} us  ratio {t12/(12*t1):.3f}")
