"""Debug: error of the split-format convolution against an fp64 reference, with and without the x_amax input scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
if os.environ.get("FF_LIB"):
    import focusflow_official_amd._hip as _h
    _h.LIB_PATH = os.environ["FF_LIB"]
from focusflow_official_amd import ops
import torch.nn.functional as F
torch.manual_seed(0)
for (cin, cout, k, hw, xmag, wmag) in [(128, 128, 3, 16, 1e-4, 0.05), (128, 128, 3, 16, 1.0, 0.05), (256, 128, 1, 16, 1e-4, 0.05), (64, 64, 3, 64, 3e-4, 0.05), (128, 128, 3, 16, 1.0, 0.002)]:
    x = (torch.randn(1, hw, hw, cin) * xmag)
    w = torch.randn(cout, cin, k, k) * wmag
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=k // 2).permute(0, 2, 3, 1)
    wp = torch.zeros(cout, k * k * cin, device="cuda")
    ops.pack_conv_weight(w.cuda(), wp, cin)
    wps = ops.pack_split(wp)
    xc = x.cuda()
    for use_amax in (False, True):
        amax = None
        if use_amax:
            amax = xc.abs().max().view(1).view(torch.int32).clone()
        y = ops.conv2d([xc], wps, None, cout, k, k, 1, (k // 2, k // 2), w_fmt=1, x_amax=amax)
        err = (y.cpu().double() - ref).abs().max().item()
        print("cin %d cout %d k%d xmag %.0e wmag %.0e amax=%d: max err %.3e  rel-to-max %.3e" % (cin, cout, k, xmag, wmag, use_amax, err, err / ref.abs().max().item()))
