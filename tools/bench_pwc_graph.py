"""FF-PWC forward (BASELINE configs[3]: 1 pair 448x1024): eager launches against replay from a captured hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from focusflow_official_amd.pwcnet import FF_PWCNET
cfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
torch.manual_seed(0)
m = FF_PWCNET(cfg).cuda().eval()
with torch.no_grad():
    m.netExtractor.netOne[0].weight.mul_(1 / 255.0); m.netExtractor.mask_netOne[0].weight.mul_(1 / 255.0)
b = int(os.environ.get("B", 1))
g = torch.Generator().manual_seed(0)
i1 = torch.randint(0, 256, (b, 3, 448, 1024), generator=g).float().cuda()
i2 = torch.roll(i1, (3, -5), (2, 3))
m1 = ((torch.rand(b, 1, 448, 1024, generator=g) < 2000 / (448 * 1024)).float() * 255).cuda()


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, o


with torch.no_grad():
    dt, ref = timeit(lambda: m(i1, i2, m1, m1, test_mode=True))
    print(f"eager : {dt * 1e3:.2f} ms  {b / dt:.1f} pairs/s")
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            m(i1, i2, m1, m1, test_mode=True)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = m(i1, i2, m1, m1, test_mode=True)

    def rep():
        gr.replay()
        return out
    dt2, o2 = timeit(rep)
    print(f"graph : {dt2 * 1e3:.2f} ms  {b / dt2:.1f} pairs/s   max|graph - eager| = {(o2 - ref).abs().max().item():.3e}")
