"""Micro-benchmark of the CorrBlock kernels at BASELINE config-2 shapes (B=8, 48x64) or config 5 (H=68 W=120):
ff_corr_build and ff_corr_lookup_tiled_fwd, fp32 or fp16 pyramid (HALF=1).  Also the target of the rocprofv3 --pmc passes
behind profiles/r02_lookup_traffic.json (ONLY=lookup keeps other kernels out of the counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from focusflow_official_amd import ops

b, h, w = int(os.environ.get("B", 8)), int(os.environ.get("H", 48)), int(os.environ.get("W", 64))
half = os.environ.get("HALF", "0") == "1"
jitter = float(os.environ.get("JITTER", 8))
g = torch.Generator().manual_seed(0)
f12 = torch.randn(2 * b, h, w, 256, generator=g).cuda()
f1, f2 = f12[:b], f12[b:]
pyr = ops.corr_build(f1, f2, half)
coords = ops.coords_init(b, h, w, f1)
coords += (torch.rand(coords.shape, generator=g) * 2 * jitter - jitter).cuda()   # test plumbing only
per_q = 2104 if half else 2904
vol_bytes = sum(lv.numel() * lv.element_size() for lv in pyr.levels)
print(f"B={b} {h}x{w} pyramid {'fp16' if half else 'fp32'}: {vol_bytes / 1e6:.1f} MB")


def timeit(fn, spoil, n=30):
    junk = torch.empty(128 * 1024 * 1024, device="cuda") if spoil else None
    ts = []
    for i in range(n):
        if spoil:
            junk.fill_(float(i))
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
    ts = sorted(ts[5:])
    return ts[len(ts) // 2], ts[0]


only = os.environ.get("ONLY", "")
for name, spoil in (("cache-warm (back-to-back)", False), ("cache-cold (512 MB written between launches)", True)):
    if only in ("", "lookup"):
        us, mn = timeit(lambda: ops.corr_lookup_tiled(pyr, coords), spoil)
        nbytes = per_q * b * h * w
        print(f"lookup {name}: median {us:.1f} us  min {mn:.1f} us -> {nbytes / us / 1e3:.0f} GB/s algorithmic = "
              f"{nbytes / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s")
    if only in ("", "build"):
        us, mn = timeit(lambda: ops.corr_build(f1, f2, half), spoil, n=15)
        flop = 2.0 * b * (h * w) ** 2 * 256
        print(f"build (incl. operand split) {name}: median {us:.1f} us  min {mn:.1f} us -> {3 * flop / us / 1e6:.0f} TFLOP/s f16 issued, "
              f"{vol_bytes / us / 1e3:.0f} GB/s written")
if only in ("", "build"):
    both = torch.empty((2 * b * h * w, 1024), dtype=torch.uint8, device="cuda")
    from focusflow_official_amd import _hip
    us, mn = timeit(lambda: _hip.call("ff_pack_split_f16", ops._p(f12), ops._p(both), 2 * b * h * w, 256, ops._stream()), False, n=15)
    print(f"operand split alone ({2 * b * h * w} rows): median {us:.1f} us")
