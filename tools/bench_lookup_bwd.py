import os, sys, torch
sys.path.insert(0, os.getcwd())
from focusflow_official_amd import ops
b, h, w = 8, 46, 62
g = torch.Generator().manual_seed(0)
f12 = torch.randn(2 * b, h, w, 256, generator=g).cuda()
pyr = ops.corr_build(f12[:b].contiguous(), f12[b:].contiguous())
coords = ops.coords_init(b, h, w, f12); coords += (torch.rand(coords.shape, generator=g) * 8 - 4).cuda()
dout = torch.randn(b, h, w, 324, generator=g).cuda()
dp = ops.TiledPyramid.empty(b * h * w, h, w, False, f12.device, zero=True)
for _ in range(3): ops.corr_lookup_tiled_bwd(dp, coords, dout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.corr_lookup_tiled_bwd(dp, coords, dout)
e1.record(); torch.cuda.synchronize()
print(f"lookup backward {b}x{h}x{w}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
# all twelve iterations in one launch (+ the pooling chain), against twelve launches + the pooling pass + the zero fill
cl = [coords + (torch.rand(coords.shape, generator=g) - 0.5).cuda() * t for t in range(12)]
dl = [torch.randn(b, h, w, 324, generator=g).cuda() for _ in range(12)]
for _ in range(2): d0 = ops.corr_lookup_tiled_bwd_all(cl, dl, h, w)
torch.cuda.synchronize()
e0.record()
for _ in range(10): d0 = ops.corr_lookup_tiled_bwd_all(cl, dl, h, w)
e1.record(); torch.cuda.synchronize()
print(f"12 lookups in one launch: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
e0.record()
for _ in range(10):
    dp = ops.TiledPyramid.empty(b * h * w, h, w, False, f12.device, zero=True)
    for c, d in zip(cl, dl): ops.corr_lookup_tiled_bwd(dp, c, d)
    ops.corr_pyramid_tiled_bwd(dp)
e1.record(); torch.cuda.synchronize()
print(f"zero fill + 12 launches + pooling pass: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
print("max |difference| of d(volume):", float((d0 - dp.levels[0]).abs().max()), "of", float(d0.abs().max()))
for T in (1, 2, 4, 8, 12):
    for _ in range(2): ops.corr_lookup_tiled_bwd_all(cl[:T], dl[:T], h, w)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): ops.corr_lookup_tiled_bwd_all(cl[:T], dl[:T], h, w)
    e1.record(); torch.cuda.synchronize()
    print(f"  T = {T:2d}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us")
