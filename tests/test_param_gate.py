"""Host logic of fn.GraphScope / fn.ParamGate (CPU, no HIP call): the weight gradients that the applications of a
shared conv accumulate in the scope's buffer must reach the parameters THROUGH autograd - exactly one hook call per
parameter and backward pass (what DDP's reducer with find_unused_parameters=False needs, common.py:49 of the reference),
also when the loss reaches only some applications, and again on a second backward over a retained graph."""
import torch
import torch.nn as nn

from focusflow_official_amd import fn


class _FakeGroup:
    """Stands in for cce.PackedConv: one conv, 'packed' gradient rows = the flat weight gradient."""

    def __init__(self, conv):
        self.convs = [conv]

    def unpack_wgrad(self, dwp, j, off):
        return dwp.view_as(self.convs[j].weight).clone()


class _App(torch.autograd.Function):
    """Same contract as fn.ConvFn inside a scope: add the parameter gradients into scope.acc[pc], return None for them."""

    @staticmethod
    def forward(ctx, scope, pc, x, w, b):
        ctx.scope, ctx.pc = scope, pc
        ctx.save_for_backward(x)
        return x * pc.convs[0].weight.detach().sum() + pc.convs[0].bias.detach().sum()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pc = ctx.pc
        acc = ctx.scope.acc.get(pc)
        if acc is None:
            acc = ctx.scope.acc[pc] = (torch.zeros(pc.convs[0].weight.numel()), torch.zeros(1))
        acc[0].add_((dy * x).sum())
        acc[1].add_(dy.sum())
        return None, None, dy * pc.convs[0].weight.detach().sum(), None, None


def _chain(n=3):
    conv = nn.Conv2d(1, 1, 1)
    pc = _FakeGroup(conv)
    calls = []
    conv.weight.register_hook(lambda g: calls.append("w"))
    conv.bias.register_hook(lambda g: calls.append("b"))
    scope = fn.GraphScope()
    x = torch.tensor([2.0], requires_grad=True)
    ys, h = [], x
    for _ in range(n):
        w, b = scope.gated(pc, [conv.weight, conv.bias])
        h = _App.apply(scope, pc, h, w, b)
        ys.append(h)
    assert len(scope.gates) == 1, "one gate per conv group and pass"
    return conv, calls, x, ys


def _reference(conv, n_reached, x0=2.0):
    w = conv.weight.detach().clone().requires_grad_(True)
    b = conv.bias.detach().clone().requires_grad_(True)
    h = torch.tensor([x0])
    for _ in range(n_reached):
        h = h * w.sum() + b.sum()
    h.sum().backward()
    return w.grad, b.grad


def test_partial_loss_delivers_one_gradient_per_parameter_through_autograd():
    conv, calls, x, ys = _chain()
    ys[0].sum().backward(retain_graph=True)          # the loss reaches the first application only
    assert sorted(calls) == ["b", "w"]
    gw, gb = _reference(conv, 1)
    assert torch.allclose(conv.weight.grad, gw) and torch.allclose(conv.bias.grad, gb)


def test_full_loss_and_second_backward_over_a_retained_graph():
    conv, calls, x, ys = _chain()
    ys[2].sum().backward(retain_graph=True)
    assert sorted(calls) == ["b", "w"]
    gw, gb = _reference(conv, 3)
    assert torch.allclose(conv.weight.grad, gw, rtol=1e-5) and torch.allclose(conv.bias.grad, gb, rtol=1e-5)
    first = conv.weight.grad.clone()
    conv.weight.grad = None
    conv.bias.grad = None
    calls.clear()
    ys[2].sum().backward()                            # starts from an empty buffer again
    assert sorted(calls) == ["b", "w"] and torch.equal(first, conv.weight.grad)


def test_gate_skips_absent_bias():
    conv = nn.Conv2d(1, 1, 1, bias=False)
    pc = _FakeGroup(conv)
    scope = fn.GraphScope()
    w, b = scope.gated(pc, [conv.weight, None])
    assert b is None and w.requires_grad and w.data_ptr() == conv.weight.data_ptr()
