import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F, copy
from focusflow_official_amd import cce, fn, ops
DEV='cuda:0'
def nhwc(t): return t.detach().permute(0,2,3,1).contiguous().to(DEV)
def nchw(t): return t.detach().cpu().permute(0,3,1,2).contiguous()
g = torch.Generator().manual_seed(1)
b,h,w=8,96,128
x = torch.randn(b,64,h,w,generator=g, requires_grad=True)
cv = nn.Conv2d(64,64,3,padding=1)
y = cv(x); gy = torch.randn(y.shape, generator=g); y.backward(gy)
dcv = copy.deepcopy(cv).to(DEV); dcv.weight.grad=None; dcv.bias.grad=None
pc = cce.PackedConv([dcv])
xd = nhwc(x).requires_grad_(True)
out = fn.conv(pc, xd)
print('fwd err', (nchw(out)-y.detach()).abs().max().item())
out.backward(nhwc(gy)); torch.cuda.synchronize()
err = (nchw(xd.grad)-x.grad).abs()
print('dx err max', err.max().item(), 'frac bad', (err>1e-3).float().mean().item())
print('per-batch max', err.amax(dim=(1,2,3)))
print('per-row (h) bad frac', (err>1e-3).float().mean(dim=(0,1,3))[:12], (err>1e-3).float().mean(dim=(0,1,3))[-6:])
print('per-col (w) bad frac', (err>1e-3).float().mean(dim=(0,1,2))[:8])
print('per-chan bad frac', (err>1e-3).float().mean(dim=(0,2,3))[:16])
print('dW err', (dcv.weight.grad.cpu()-cv.weight.grad).abs().max().item(), cv.weight.grad.abs().max().item())
# run dgrad twice to see determinism
xd2 = nhwc(x).requires_grad_(True); out2 = fn.conv(pc, xd2); out2.backward(nhwc(gy)); torch.cuda.synchronize()
print('dx run-to-run diff', (xd2.grad-xd.grad).abs().max().item())
