"""The discriminator's fused resampling / activation glue (csrc/pool_up.hip, ops/pool_up.py; reference: the LeakyReLU / nn.Upsample /
nn.AvgPool2d modules and the residual sum of lib/discriminators/unet_discriminators.py:8-72) against torch's own operators in
float64: values, gradients, and the gradient of a gradient (what the R1 penalty asks of every layer of the discriminator)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
pu = importlib.import_module("3dhumangan_amd.lib.components.ops.pool_up")
DEV = "cuda"

CASES = {
    "lrelu_up2": (lambda x: pu.lrelu_up2(x, 0.2), lambda x: F.interpolate(F.leaky_relu(x, 0.2), scale_factor=2, mode="nearest"), 1, False),
    "up2": (pu.up2, lambda x: F.interpolate(x, scale_factor=2, mode="nearest"), 1, False),
    "up2_add": (pu.up2_add, lambda s, d: F.interpolate(s, scale_factor=2, mode="nearest") + d, 2, True),
    "avgpool2": (pu.avgpool2, lambda x: F.avg_pool2d(x, 2), 1, False),
    "avgpool2_sum": (pu.avgpool2_sum, lambda s, d: F.avg_pool2d(s + d, 2), 2, False),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("B,C,H,W,dtype", [(2, 64, 8, 6, torch.float32), (1, 3, 4, 10, torch.float32), (2, 128, 16, 8, torch.float16),
                                           (1, 20, 6, 4, torch.float16), (1, 256, 32, 16, torch.float32)])
def test_against_torch_up_to_second_order(name, B, C, H, W, dtype):
    fn, ref_fn, n_in, second_is_big = CASES[name]
    g = torch.Generator().manual_seed(C + H)
    shapes = [(B, C, H, W)] * n_in
    if second_is_big:
        shapes[1] = (B, C, 2 * H, 2 * W)
    ins = [torch.randn(s, generator=g).to(dtype) for s in shapes]
    ref_in = [t.double().requires_grad_() for t in ins]
    ref = ref_fn(*ref_in)
    cot = torch.randn(ref.shape, generator=g).to(dtype)
    dev_in = [t.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for t in ins]
    out = fn(*dev_in)
    tol = 2e-6 if dtype == torch.float32 else 2e-3
    assert out.dtype == dtype and out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) < tol
    # first order, with a graph: the penalty below differentiates it again
    cd = cot.to(DEV).requires_grad_()
    cr = cot.double().requires_grad_()
    grads = torch.autograd.grad(out, dev_in, cd, create_graph=True)
    rgrads = torch.autograd.grad(ref, ref_in, cr, create_graph=True)
    for a, b in zip(grads, rgrads):
        assert rel_err(a.detach().cpu(), b.detach()) < tol
    # second order: a scalar function of the gradients, differentiated with respect to the cotangent (the gradient's own inputs)
    pen = sum((a.float() ** 2).sum() for a in grads)
    rpen = sum((b ** 2).sum() for b in rgrads)
    (gc,) = torch.autograd.grad(pen, cd)
    (rgc,) = torch.autograd.grad(rpen, cr)
    assert rel_err(gc.cpu(), rgc) < (1e-5 if dtype == torch.float32 else 5e-3)


def test_non_channels_last_and_sliced_inputs_are_accepted():
    x = torch.randn(2, 64, 8, 8, device=DEV)                                   # NCHW-contiguous
    wide = torch.randn(2, 96, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    assert rel_err(pu.avgpool2(x).cpu(), F.avg_pool2d(x, 2).cpu()) < 2e-6
    assert rel_err(pu.lrelu_up2(wide[:, 16:80]).cpu(), F.interpolate(F.leaky_relu(wide[:, 16:80], 0.2), scale_factor=2).cpu()) < 2e-6
    with pytest.raises(ValueError):
        pu.avgpool2(torch.randn(1, 4, 5, 4, device=DEV))


@pytest.mark.parametrize("co,ci,k", [(128, 3, 3), (512, 512, 3), (64, 256, 1), (256, 1024, 1), (26, 64, 1)])
def test_fused_spectral_norm_matches_torchs_hook(co, ci, k):
    """ops/spectral.py (h3d_spectral_norm / _bwd) against torch.nn.utils.spectral_norm in float64: the normalised weight, the
    updated u / v buffers over two forwards (the GAN pattern: D(real), D(fake), one backward through both), and the gradient
    with respect to weight_orig."""
    disc = importlib.import_module("3dhumangan_amd.lib.discriminators.unet_discriminators")
    torch.manual_seed(co + ci)
    m = disc._conv(ci, co, k, True).to(DEV).train()
    assert hasattr(m, "_sn_hook") and not any(type(h).__name__ == "SpectralNorm" for h in m._forward_pre_hooks.values())
    ref = torch.nn.utils.spectral_norm(torch.nn.Conv2d(ci, co, k, 1, k // 2)).double()
    ref.load_state_dict({n: t.detach().cpu().double() for n, t in m.state_dict().items()})
    ref.train()
    x = torch.randn(2, ci, 8, 8)
    cots = [torch.randn(2, co, 8, 8) for _ in range(2)]
    loss = rloss = 0
    for c in cots:
        loss = loss + (torch.nn.functional.conv2d(x.to(DEV), _weight_after_forward(m, x.to(DEV)), m.bias, padding=k // 2) * c.to(DEV)).sum()
        rloss = rloss + (ref(x.double()) * c.double()).sum()
        assert rel_err(m.weight.detach().cpu(), ref.weight.detach()) < 2e-6
        assert rel_err(m.weight_u.cpu(), ref.weight_u) < 2e-6 and rel_err(m.weight_v.cpu(), ref.weight_v) < 2e-6
    loss.backward()
    rloss.backward()
    assert rel_err(m.weight_orig.grad.cpu(), ref.weight_orig.grad) < 1e-5
    # eval mode goes through torch's own compute_weight (no power iteration): same value as the reference module in eval mode
    m.eval(), ref.eval()
    _weight_after_forward(m, x.to(DEV))
    ref(x.double())
    assert rel_err(m.weight.detach().cpu(), ref.weight.detach()) < 2e-6


def _weight_after_forward(m, x):
    m(x)                          # Conv2d.forward sets m.weight (fused spectral normalisation) before convolving
    return m.weight
