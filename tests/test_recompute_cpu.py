"""Host logic of ops/recompute.py (gradients of forward-only kernels by recomputation) on the CPU: the kernel launch is replaced by
a stand-in that computes the same function without recording a graph."""
import importlib

import torch

rc = importlib.import_module("3dhumangan_amd.lib.components.ops.recompute")


def _restate(x, w, b):
    return torch.tanh(x @ w.t()) * b


def _kernel(x, w, b):                      # what a HIP launch looks like to autograd: values only
    assert not torch.is_grad_enabled()
    return _restate(x.detach(), w.detach(), b.detach())


def test_gradients_match_the_restatement_and_only_for_inputs_that_need_them():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 7, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 7, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, generator=g, dtype=torch.float64)                     # no gradient wanted
    cot = torch.randn(5, 3, generator=g, dtype=torch.float64)
    y = rc.with_recomputed_grad(_kernel, _restate, x, w, b)
    assert y.requires_grad and torch.equal(y.detach(), _restate(x, w, b).detach())
    gx, gw = torch.autograd.grad(y, [x, w], cot)
    rx, rw = torch.autograd.grad(_restate(x, w, b), [x, w], cot)
    assert torch.allclose(gx, rx, atol=1e-14) and torch.allclose(gw, rw, atol=1e-14)
    # a single input
    y = rc.with_recomputed_grad(_kernel, _restate, x.detach(), w, b)
    (gw2,) = torch.autograd.grad(y, [w], cot)
    assert torch.allclose(gw2, rw, atol=1e-14)


def test_no_graph_when_nothing_needs_a_gradient_or_grad_is_disabled():
    x, w, b = torch.randn(2, 4), torch.randn(3, 4), torch.randn(3)
    assert not rc.with_recomputed_grad(_kernel, _restate, x, w, b).requires_grad
    w.requires_grad_()
    with torch.no_grad():
        assert not rc.with_recomputed_grad(_kernel, _restate, x, w, b).requires_grad
    assert rc.with_recomputed_grad(_kernel, _restate, x, w, b).requires_grad
