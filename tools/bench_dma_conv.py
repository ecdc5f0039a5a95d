"""Per-layer timing of the update block's convolutions at the headline shape (8 pairs: 8 x 48 x 64 pixels): the fp32-input
route (conv_patch.hip) against conv_dma.hip over split-pair inputs, for every tile shape of the latter.
    python tools/bench_dma_conv.py [B]
FF_DMA_TILE is a tuning override read per call (this tool sets it); bench.py refuses to run with it set."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focusflow_official_amd import ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = (int(v) for v in os.environ.get("HW", "48x64").split("x"))
# (name, segments, cout, kh, kw, epilogue)
LAYERS = [
    ("convc2 256->192 3x3", [256], 192, 3, 3, None),
    ("convf2 128->64 3x3", [128], 64, 3, 3, None),
    ("conv 192+64->126 3x3", [192, 64], 126, 3, 3, None),
    ("zr 128+128->256 1x5", [128, 128], 256, 1, 5, "rh"),
    ("q 128+128->128 1x5", [128, 128], 128, 1, 5, "blend"),
    ("zr 128+128->256 5x1", [128, 128], 256, 5, 1, "rh"),
    ("q 128+128->128 5x1", [128, 128], 128, 5, 1, "blend"),
    ("heads 128->512 3x3", [128], 512, 3, 3, None),
]


if os.environ.get("ENC"):       # the eval-BatchNorm encoder's residual-block layers (HW=192x256 / 96x128 / 48x64 with B = 8)
    c = int(os.environ["ENC"])
    LAYERS = [(f"enc {c}->{c} 3x3", [c], c, 3, 3, None)]
if os.environ.get("EXP"):       # fixed cost vs slope: the z|r layer over 1, 2, 3 input segments (K = 640, 1280, 1920)
    LAYERS = [("zr K640 1x5", [128], 256, 1, 5, "rh"), ("zr K1280 1x5", [128, 128], 256, 1, 5, "rh"), ("zr K1920 1x5", [128, 128, 128], 256, 1, 5, "rh"),
              ("plain K640 1x5", [128], 256, 1, 5, None), ("plain K1280 1x5", [128, 128], 256, 1, 5, None), ("plain K1920 1x5", [128, 128, 128], 256, 1, 5, None)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    tot_old = tot_best = 0.0
    only = os.environ.get("LAYER")            # substring filter (profiling runs)
    tiles = [int(t) for t in os.environ.get("TILES", "0,8,8,4").split(",")]
    for name, segs, cout, kh, kw, ep in LAYERS:
        if only and only not in name:
            continue
        cin = sum(segs)
        xs = [torch.randn(B, H, W, c, generator=g).to(DEV) for c in segs]
        sp = [ops.split_copy(x) for x in xs]
        wt = (torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5).to(DEV)
        wp = torch.empty(cout, kh * kw * cin, device=DEV)
        ops.pack_conv_weight(wt, wp, cin)
        wp = ops.pack_split(wp)
        wf = ops.pack_frag16(wp, cout) if os.environ.get('FRAG', '1') != '0' else None
        bias = torch.randn(cout, generator=g).to(DEV)
        res = torch.randn(B, H, W, cout, generator=g).to(DEV)
        hprev = torch.randn(B, H, W, 128, generator=g).to(DEV)
        z = torch.rand(B, H, W, 128, generator=g).to(DEV)
        pad = (kh // 2, kw // 2)
        kw_old, kw_new = {}, {}
        if ep == "rh":
            kw_old = dict(res=res, act_res=2, ep_rh=hprev, ep_split=128)
            kw_new = dict(kw_old, y_split=128)
        elif ep == "blend":
            kw_old = dict(res=res, act_res=3, ep_blend=(z, hprev))
            kw_new = dict(kw_old, y2_split=True)
        else:
            kw_old = dict(act=1)
            kw_new = dict(act=1, y_split=cout % 32 == 0)
        out_o = torch.empty(B, H, W, (cout + 3) // 4 * 4, device=DEV)
        out_n = torch.empty(B, H, W, (cout + 31) // 32 * 32, device=DEV)
        os.environ.pop("FF_DMA_TILE", None)
        t_old = 0.0 if os.environ.get("NO_OLD") else timeit(lambda: ops.conv2d(xs, wp, bias, cout, kh, kw, 1, pad, w_fmt=1, out=out_o[..., :cout], **kw_old))
        flop = 2.0 * B * H * W * cout * cin * kh * kw
        line = f"{name:24s} old {t_old:6.1f} us ({flop / max(t_old, 1e-9) / 1e6:5.0f} TF/s)  dma:"
        best = 1e9
        for tile in tiles:
            if tile:
                os.environ["FF_DMA_TILE"] = str(tile)
            else:
                os.environ.pop("FF_DMA_TILE", None)
            t = timeit(lambda: ops.conv2d(sp, wp, bias, cout, kh, kw, 1, pad, w_fmt=1, out=out_n[..., :cout], w_frag=wf, **kw_new))
            line += f"  {'auto' if not tile else tile}: {t:6.1f}"
            best = min(best, t)
        os.environ.pop("FF_DMA_TILE", None)
        print(line + f"   best {best:6.1f} us ({flop / best / 1e6:5.0f} TF/s)", flush=True)
        tot_old += t_old
        tot_best += best
    print(f"sum per iteration: old {tot_old:.1f} us, best dma {tot_best:.1f} us")


if __name__ == "__main__":
    main()
