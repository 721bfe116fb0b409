"""CPU restatements of the arithmetic behind round 4's discriminator kernels (csrc/spectral_norm.hip, csrc/pool_up.hip), against torch's
own operators -- the formulas the kernels implement, independent of any GPU.

  * spectral normalisation (reference: torch.nn.utils.spectral_norm on every convolution, lib/discriminators/unet_discriminators.py:17):
    forward  v' = normalize(W^T u), u' = normalize(W v'), sigma = u'.(W v') = |W v'| (when |W v'| > eps), W_sn = W / sigma;
    backward dW = (G - sum(G * W_sn) u' v'^T) / sigma   with u', v' constants -- what h3d_spectral_norm_bwd computes;
  * up2 / pool2 with a LeakyReLU mask are adjoint linear maps for a fixed mask (each is the other's backward), and
    up(lrelu(x)) = up2(x; mask = x), avgpool(s + d) = pool2(s + d; 1/4) -- the identities ops/pool_up.py builds on;
  * the exact reordering of ResBlock.forward: a 1x1 convolution commutes with nearest-neighbour upsampling bit for bit."""
import torch
import torch.nn.functional as F


def test_spectral_norm_backward_formula_matches_autograd():
    torch.manual_seed(0)
    m = torch.nn.utils.spectral_norm(torch.nn.Conv2d(6, 10, 3, padding=1)).double()
    m.train()
    w0, u0 = m.weight_orig.detach().clone(), m.weight_u.detach().clone()
    x = torch.randn(2, 6, 5, 5, dtype=torch.float64)
    cot = torch.randn(2, 10, 5, 5, dtype=torch.float64)
    (m(x) * cot).sum().backward()
    # the kernel's forward, restated
    W = w0.reshape(10, -1)
    t = W.t() @ u0
    v = t / t.norm().clamp_min(1e-12)
    s = W @ v
    u = s / s.norm().clamp_min(1e-12)
    sigma = torch.dot(u, s)
    assert torch.allclose(sigma, s.norm())                                   # sigma = |W v'|: what sn_scale computes
    assert torch.allclose(m.weight_u, u) and torch.allclose(m.weight_v, v)
    w_sn = (W / sigma).reshape_as(w0)
    assert torch.allclose(m.weight.detach(), w_sn)
    # G = dL/dW_sn from the convolution alone, then the closed form
    wl = w_sn.clone().requires_grad_(True)
    (F.conv2d(x, wl, m.bias.detach(), padding=1) * cot).sum().backward()
    G = wl.grad.reshape(10, -1)
    c = (G * w_sn.reshape(10, -1)).sum()
    dW = (G - c * torch.outer(u, v)) / sigma
    assert torch.allclose(dW.reshape_as(w0), m.weight_orig.grad, rtol=1e-10, atol=1e-12)


def _up2(x, mask=None, slope=0.2, scale=1.0, addend=None):
    m = 1.0 if mask is None else torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))
    y = F.interpolate(scale * m * x, scale_factor=2, mode="nearest")
    return y if addend is None else y + addend


def _pool2(x, x2=None, mask=None, slope=0.2, scale=1.0):
    s = x if x2 is None else x + x2
    p = F.avg_pool2d(s, 2) * 4.0
    m = 1.0 if mask is None else torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))
    return scale * m * p


def test_up2_and_pool2_are_adjoint_for_a_fixed_mask():
    torch.manual_seed(1)
    x = torch.randn(2, 5, 4, 6, dtype=torch.float64)
    y = torch.randn(2, 5, 8, 12, dtype=torch.float64)
    mask = torch.randn(2, 5, 4, 6, dtype=torch.float64)
    for mk in (None, mask):
        lhs = (_up2(x, mk, 0.2, 0.7) * y).sum()
        rhs = (x * _pool2(y, None, mk, 0.2, 0.7)).sum()
        assert torch.allclose(lhs, rhs)
    # the identities the discriminator uses
    assert torch.equal(_up2(x, mask=x), F.interpolate(F.leaky_relu(x, 0.2), scale_factor=2, mode="nearest"))
    s, d = torch.randn_like(y), torch.randn_like(y)
    assert torch.allclose(_pool2(s, d, None, 1.0, 0.25), F.avg_pool2d(s + d, 2))
    assert torch.allclose(F.avg_pool2d(s, 2) + F.avg_pool2d(d, 2), F.avg_pool2d(s + d, 2))       # one pooling of the sum
    # and autograd agrees that pool2(.; mask = x) is the backward of up(lrelu(x))
    xr = x.clone().requires_grad_(True)
    (F.interpolate(F.leaky_relu(xr, 0.2), scale_factor=2, mode="nearest") * y).sum().backward()
    assert torch.allclose(xr.grad, _pool2(y, None, x, 0.2, 1.0))


def test_one_by_one_convolution_commutes_with_nearest_upsampling_bit_for_bit():
    torch.manual_seed(2)
    x = torch.randn(2, 16, 6, 5)
    w, b = torch.randn(8, 16, 1, 1), torch.randn(8)
    a = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b)          # the reference's order (ResBlock.shortcut)
    c = F.interpolate(F.conv2d(x, w, b), scale_factor=2, mode="nearest")          # this build's order
    assert torch.equal(a, c)
