"""The AMP tier (reference: torch.cuda.amp.autocast + GradScaler around both networks, lib/trainers/base_trainer.py:50-51,
lib/trainers/phase_trainer.py:270-283): under float16 autocast activations and their gradients travel as f16 through the native kernels
(h3d_conv_x3_f16, h3d_conv_wgrad_x3_f16, h3d_wgrad_x3_bias_f16, h3d_*spade*_f16), weights stay fp32, accumulation is fp32.
Checked against the REFERENCE module under float16 autocast (tests/golden/disc_tiny_amp.npz, generated from /root/reference by
tests/golden/make_golden_train.py) -- not against the product's own fp32 run -- and against float64 on f16-exact inputs."""
import importlib
import json
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_golden, rel_err

pytestmark = pytest.mark.gpu
conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
spade = importlib.import_module("3dhumangan_amd.lib.components.ops.spade")
disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
losses = importlib.import_module("3dhumangan_amd.lib.trainers.losses")
DEV = "cuda"


@pytest.mark.parametrize("B,H,W,ci,co,k", [(2, 16, 16, 128, 128, 3), (1, 9, 7, 64, 256, 1), (2, 32, 16, 256, 128, 3), (1, 8, 8, 3, 128, 3)])
def test_conv_f16_forward_and_gradients_vs_float64(B, H, W, ci, co, k):
    torch.manual_seed(ci + co + k)
    x = torch.randn(B, ci, H, W).half()
    w = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    b = torch.randn(co) * 0.1
    cot = torch.randn(B, co, H, W).half()
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.conv2d(xd, wd, bd, padding=k // 2)
    gx, gw, gb = torch.autograd.grad(ref, [xd, wd, bd], cot.double())
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    wg, bg = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    assert conv.supported(xg, wg)
    out = conv.conv2d(xg, wg, bg)
    assert out.dtype == torch.float16
    assert rel_err(out.detach().cpu(), ref.detach()) < 1.5e-3            # one rounding to f16 at the store (2^-11)
    hx, hw, hb = torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV))
    assert hx.dtype == torch.float16 and hw.dtype == torch.float32 and hb.dtype == torch.float32
    assert rel_err(hx.cpu(), gx) < 1.5e-3
    assert rel_err(hw.cpu(), gw) < 1e-4                                  # f16 operands are exact in the bf16 hi/lo split, fp32 result
    assert rel_err(hb.cpu(), gb) < 1e-5


@pytest.mark.parametrize("B,H,W,ci,co,k", [(2, 16, 16, 128, 128, 3), (1, 9, 7, 64, 256, 1), (2, 32, 16, 256, 512, 3), (1, 1, 4099, 256, 256, 1)])
def test_conv_f16_single_weight_plane_is_the_autocast_arithmetic(B, H, W, ci, co, k, monkeypatch):
    """Round 5: the AMP tier's default rounds the weight to f16 ONCE (h3d_conv_x3_f16x1: one product per weight, a ring stage carries
    two k-steps) -- what float16 autocast computes in the reference (lib/trainers/base_trainer.py:50-51).  Against float64 on the
    f16-ROUNDED weights only the output's own rounding to f16 is left; against the two-plane tier the difference is the weights' 11
    bits."""
    torch.manual_seed(ci + co + k)
    x = torch.randn(B, ci, H, W).half()
    w = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    b = torch.randn(co) * 0.1
    cot = torch.randn(B, co, H, W).half()
    xd, wd = x.double().requires_grad_(), w.half().double().requires_grad_()
    ref = F.conv2d(xd, wd, b.double(), padding=k // 2)
    gx, = torch.autograd.grad(ref, [xd], cot.double())
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    wg, bg = w.to(DEV).requires_grad_(), b.to(DEV)
    outs = {}
    for planes in (1, 2):
        monkeypatch.setattr(conv, "AMP_WEIGHT_PLANES", planes)
        out = conv.conv2d(xg, wg, bg)
        hx, hw = torch.autograd.grad(out, [xg, wg], cot.to(DEV))
        outs[planes] = (out.detach().float().cpu(), hx.float().cpu(), hw.cpu())
    assert conv.pack_stream(wg.detach(), half=True, planes=1).numel() * 2 == conv.pack_stream(wg.detach(), half=True, planes=2).numel()
    assert rel_err(outs[1][0], ref.detach()) < 6e-4                     # only the f16 rounding of the result (2^-11 of its magnitude)
    assert rel_err(outs[1][1], gx) < 6e-4                               # backward-data: the same kernel on the transposed stream
    assert 1e-6 < rel_err(outs[1][0], outs[2][0]) < 1.5e-3              # differs from the two-plane tier by the weights' rounding
    assert torch.equal(outs[1][2], outs[2][2])                          # the weight gradient never reads the weight stream


def test_f16_matrix_products_keep_subnormal_operands():
    """The AMP convolution / weight-gradient kernels multiply the f16 values themselves on v_mfma_f32_32x32x16_f16 (round 4: one exact
    plane, no bf16 split).  f16 subnormals (|x| < 6.1e-5: small activations, unscaled gradients) must take part in the products
    like any other value -- operands entirely in the subnormal range against float64."""
    torch.manual_seed(3)
    B, H, W, ci, co = 2, 16, 16, 128, 128
    x = (torch.randn(B, ci, H, W) * 2e-5).half()                   # every element subnormal or nearly so
    assert float((x.abs() < 6.1e-5).float().mean()) > 0.95
    w = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    cot = (torch.randn(B, co, H, W) * 2e-5).half()
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv2d(xd, wd, padding=1)
    gx, gw = torch.autograd.grad(ref, [xd, wd], cot.double())
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    wg = w.to(DEV).requires_grad_()
    out = conv.conv2d(xg, wg)
    hx, hw = torch.autograd.grad(out, [xg, wg], cot.to(DEV))
    # the outputs are themselves subnormal f16 (spacing 6e-8): compare in units of the reference's range
    assert rel_err(out.detach().float().cpu(), ref.detach()) < 2e-2
    assert rel_err(hw.cpu(), gw) < 1e-4                            # fp32 result of exact products: the subnormals were not flushed
    assert rel_err(hx.float().cpu(), gx) < 5e-2


def test_spade_kernels_f16_vs_the_fp32_kernels():
    from _torch_spade_kernels import TorchKernels
    torch.manual_seed(0)
    B, P, C = 2, 700, 64
    x = torch.randn(B, P, C).half()
    dy = torch.randn(B, P, C).half()
    for pix in (True, False):
        gamma = (torch.randn(B, P, C) * 0.3).half() if pix else torch.randn(B, C) * 0.3
        beta = (torch.randn(B, P, C) * 0.3).half() if pix else torch.randn(B, C) * 0.3
        mean, rstd = torch.randn(C) * 0.1, torch.rand(C) + 0.5
        g, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
        tk, hk = TorchKernels(), spade.HipKernels()
        f = lambda t: t.float()
        scale, shift = rstd * g, b - mean * rstd * g
        y_ref = tk.forward(f(x), scale, shift, f(gamma), f(beta))
        y = hk.forward(x.to(DEV), scale.to(DEV), shift.to(DEV), gamma.to(DEV), beta.to(DEV))
        assert y.dtype == torch.float16 and rel_err(y.cpu(), y_ref) < 1.5e-3
        m_ref = tk.moments(f(x))
        assert rel_err(hk.moments(x.to(DEV)).cpu(), m_ref) < 1e-5
        s_ref = tk.backward_sums(f(x), mean, rstd, g, b, f(gamma), f(beta), f(dy))
        s = hk.backward_sums(x.to(DEV), mean.to(DEV), rstd.to(DEV), g.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), dy.to(DEV))
        assert rel_err(s.cpu(), s_ref) < 1e-4
        c1, c2 = torch.randn(C) * 0.01, torch.randn(C) * 0.01
        ref = tk.backward_apply(f(x), mean, rstd, g, b, f(gamma), f(beta), f(dy), c1, c2)
        got = hk.backward_apply(x.to(DEV), mean.to(DEV), rstd.to(DEV), g.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), dy.to(DEV),
                                c1.to(DEV), c2.to(DEV))
        assert got[0].dtype == torch.float16
        for a, r in zip(got, ref):
            assert rel_err(a.float().cpu(), r) < 2e-3


@pytest.mark.parametrize("ci,co", [(128, 256), (256, 3), (3, 256), (256, 1)])
def test_dense_layer_under_autocast_vs_float64(ci, co):
    """wide layers: h3d_wgrad_x3_bias_f16; ToRGB / head / coordinate layers (one side <= 4): h3d_wgrad_narrow_f16."""
    torch.manual_seed(1)
    M = 18000                              # above linear.MIN_ROWS
    x = torch.randn(3, M // 3 + 1, ci)[:, : M // 3].contiguous()
    w, b = torch.randn(co, ci) / ci ** 0.5, torch.randn(co) * 0.1
    cot = torch.randn(3, M // 3, co)
    xh = x.half()
    xd, wd, bd = xh.double().requires_grad_(), w.half().double().requires_grad_(), b.half().double().requires_grad_()
    ref = F.linear(xd, wd, bd)
    gx, gw, gb = torch.autograd.grad(ref, [xd, wd, bd], cot.half().double())
    xg, wg, bg = xh.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    with torch.autocast("cuda", dtype=torch.float16):
        out = lin.linear(xg, wg, bg)
    fn = out.grad_fn
    while fn is not None and "View" in type(fn).__name__:          # the [B, N, C] view of the [M, C] product
        fn = fn.next_functions[0][0]
    assert out.dtype == torch.float16 and type(fn).__name__.startswith("_LinearAmp"), type(fn).__name__
    assert rel_err(out.detach().cpu(), ref.detach()) < 3e-3
    hx, hw, hb = torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV).half())
    assert hw.dtype == torch.float32
    assert rel_err(hx.cpu(), gx) < 3e-3
    assert rel_err(hw.cpu(), gw) < 1e-3          # f16 operands, fp32 accumulation over 18000 rows (the reference weights here are the f16-rounded ones)
    assert rel_err(hb.cpu(), gb) < 1e-3


def test_discriminator_under_autocast_vs_the_reference_under_autocast(monkeypatch):
    """The reference's UNetDiscriminator under float16 autocast (fixture) vs this build's under float16 autocast on the GPU: the
    network stays on the native kernels (counted), its outputs, losses and a weight gradient of the whole D loss agree with the
    reference's AMP run to f16-rounding accuracy, and its distance to the fp32 truth is that of the reference's AMP run."""
    g = load_golden("disc_tiny")
    a = load_golden("disc_tiny_amp")
    info = json.load(open(os.path.join(GOLDEN, "disc_tiny.json")))
    D = disc.UNetDiscriminator(**info["kwargs"]).eval()
    D.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["state"].items()})
    D = D.to(DEV)
    calls = {"f16": 0, "f32": 0}
    orig = conv._run_conv

    def counted(x, w, bias=None, transposed=False):
        calls["f16" if x.dtype == torch.float16 else "f32"] += 1
        return orig(x, w, bias, transposed)

    monkeypatch.setattr(conv, "_run_conv", counted)
    real = g["real"].to(DEV).requires_grad_(True)
    fake, gt = g["fake"].to(DEV), g["gt_segments"].to(DEV)
    meta = info["meta"]
    with torch.autocast("cuda", dtype=torch.float16):
        o_real = D(real, None, 1.0)
        o_fake = D(fake, None, 1.0)
    assert o_real["prediction"].dtype == torch.float16
    assert calls["f16"] >= 10 and calls["f32"] == 0          # every native convolution ran on f16 activations
    o_real = {k: v.float() for k, v in o_real.items()}
    o_fake = {k: v.float() for k, v in o_fake.items()}
    for k in ("prediction", "segments"):
        e_amp = rel_err(o_real[k].detach().cpu(), a["out_real"][k])
        e_ref32 = rel_err(a["out_real"][k], g["out_real"][k])          # what f16 autocast costs the REFERENCE
        e_own32 = rel_err(o_real[k].detach().cpu(), g["out_real"][k])
        print(f"{k}: vs reference-under-autocast {e_amp:.2e}; reference AMP vs fp32 {e_ref32:.2e}; ours AMP vs fp32 {e_own32:.2e}")
        assert e_amp < 1e-2 and e_own32 < max(2.0 * e_ref32, 2e-3)
        assert rel_err(o_fake[k].detach().cpu(), a["out_fake"][k]) < 1e-2
    gan = losses.logistic_d_loss(o_real["prediction"], o_fake["prediction"], meta["gan_lambda"])
    grad = losses.r1_gradient(real, o_real, meta["gan_lambda"])
    r1 = 0.5 * meta["r1_lambda"] * losses.r1_statistic(grad, "reference").mean()
    s_real = losses.segmentation_loss(o_real["segments"], gt, meta["label_dim"])[0]
    s_gen = losses.segmentation_loss(o_fake["segments"], torch.zeros_like(gt), meta["label_dim"])[0]
    for name, got in (("gan", gan), ("r1", r1), ("seg_real", s_real), ("seg_gen", s_gen)):
        want = float(a["loss"][name])
        assert abs(float(got) - want) <= 2e-2 * abs(want) + 1e-6, (name, float(got), want)
    loss = gan + 4 * r1 + (s_real + s_gen)
    loss.backward()
    gkey = info["grad_key"]
    got = dict(D.named_parameters())[gkey].grad.cpu()
    e = rel_err(got, a["grad"][gkey])
    e32 = rel_err(a["grad"][gkey], g["grad"][gkey])
    print(f"weight gradient of the D loss: vs reference-under-autocast {e:.2e} (reference AMP vs fp32: {e32:.2e})")
    assert e < 3e-2


def test_dense_layer_in_front_of_a_spade_under_autocast(monkeypatch):
    """linear(.., add=, moments=True) under float16 autocast: the own f16 GEMM with the residual addend and the batch moments in its
    epilogue -- output against the library path (F.linear on halves + add), moments against the stored output, gradients against the
    moment-less call; with and without autograd recording (the D step's generator forward runs under no_grad)."""
    monkeypatch.setattr(lin, "AMP_FUSED_MOMENTS", True)          # opt-in (H3D_AMP_FUSED_MOMENTS=1): measured slower in the iteration, not the default
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 9000, 256, generator=g).to(DEV)
    w, b = (torch.randn(256, 256, generator=g) * 0.06).to(DEV).requires_grad_(True), torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(2, 9000, 256, generator=g).to(DEV).half()
    with torch.autocast("cuda", dtype=torch.float16):
        xx = x.clone().requires_grad_(True)
        y, partial = lin.linear(xx, w, b, add=r, moments=True)
        assert y.dtype == torch.float16 and partial is not None and partial.shape[1:] == (2, 256)
        sums = partial.double().sum(0)
        assert rel_err(sums[0], y.double().sum((0, 1))) < 1e-6 and rel_err(sums[1], (y.double() ** 2).sum((0, 1))) < 1e-6
        want = torch.nn.functional.linear(x.half().double(), w.detach().half().double(), b.detach().half().double()) + r.double()
        assert rel_err(y.double(), want) < 1e-3
        (y.float().square().sum() * 1e-3).backward()             # gradients well inside the f16 range
        gx, gw, gb = xx.grad.clone(), w.grad.clone(), b.grad.clone()
        w.grad = b.grad = None
        x2 = x.clone().requires_grad_(True)
        y2 = lin.linear(x2, w, b) + r
        (y2.float().square().sum() * 1e-3).backward()
        assert rel_err(gx, x2.grad) < 2e-3 and rel_err(gw, w.grad) < 2e-3 and rel_err(gb, b.grad) < 2e-3
        with torch.no_grad():
            y3, p3 = lin.linear(x, w, b, add=r, moments=True)
        assert torch.equal(y3, y) and torch.equal(p3, partial)


def test_dense_layer_with_a_residual_addend_under_autocast(monkeypatch):
    """linear(x, w, b, add=r) under float16 autocast on the own f16 GEMM with the addend in its epilogue (fp32 sum, one rounding; opt-in,
    H3D_AMP_LINEAR_ADD=x3): values against float64 on the rounded operands, gradients against the library path + a separate sum."""
    monkeypatch.setattr(lin, "AMP_ADD_NATIVE", True)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 9000, 256, generator=g).to(DEV)
    w, b = (torch.randn(256, 256, generator=g) * 0.06).to(DEV).requires_grad_(True), torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(2, 9000, 256, generator=g).to(DEV).half().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        xx = x.clone().requires_grad_(True)
        y = lin.linear(xx, w, b, add=r)
        assert y.dtype == torch.float16 and type(y.grad_fn).__name__.startswith("_LinearAmp")
        want = torch.nn.functional.linear(x.half().double(), w.detach().half().double(), b.detach().double()) + r.detach().double()
        assert rel_err(y.double(), want) < 1e-3
        (y.float().square().sum() * 1e-3).backward()
        got = [t.grad.clone() for t in (xx, w, b, r)]
        for t in (w, b, r):
            t.grad = None
        monkeypatch.setattr(lin, "AMP_ADD_NATIVE", False)
        x2 = x.clone().requires_grad_(True)
        y2 = lin.linear(x2, w, b, add=r)
        assert not type(y2.grad_fn).__name__.startswith("_LinearAmp")        # the default: library GEMM + a sum
        (y2.float().square().sum() * 1e-3).backward()
        for a, e in zip(got, (x2.grad, w.grad, b.grad, r.grad)):
            assert rel_err(a.float(), e.float()) < 3e-3
