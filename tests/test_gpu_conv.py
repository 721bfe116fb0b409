"""The discriminator's native convolutions (csrc/conv_x3.hip, wgrad_x3.hip through lib/components/ops/conv.py) against
torch's F.conv2d in float64 on the CPU: forward, first-order gradients, and the second-order gradients of an R1-style
penalty ||d out / d x||^2 (the double backward is a composition of the same three kernels)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
TOL = 1e-3          # split bf16 (16 mantissa bits per operand): measured ~2e-5


@pytest.mark.parametrize("B,H,W,ci,co,k", [(2, 16, 8, 128, 128, 3), (1, 9, 7, 64, 64, 3), (3, 5, 11, 256, 64, 3), (1, 8, 4, 512, 512, 3),
                                            (2, 6, 6, 1024, 256, 3), (2, 16, 8, 128, 256, 1), (1, 33, 17, 64, 128, 3),
                                            (1, 4, 2, 512, 512, 1), (2, 16, 8, 3, 128, 3), (1, 9, 5, 64, 1, 1), (2, 8, 8, 64, 26, 1),
                                            (1, 6, 6, 6, 128, 1), (2, 128, 128, 64, 128, 3), (3, 96, 128, 128, 64, 3), (1, 256, 128, 64, 64, 3),
                                            (2, 128, 128, 192, 192, 3)])
def test_conv_forward_and_first_order_gradients(B, H, W, ci, co, k):
    g = torch.Generator().manual_seed(ci + co + k)
    x = torch.randn(B, ci, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(co, ci, k, k, generator=g, dtype=torch.float64) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g, dtype=torch.float64)
    proj = torch.randn(B, co, H, W, generator=g, dtype=torch.float64)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=k // 2)
    (ref * proj).sum().backward()
    xd, wd, bd = (t.float().to(DEV).requires_grad_(True) for t in (x, w, b))
    assert conv.supported(xd, wd)
    got = conv.conv2d(xd, wd, bd)
    assert got.shape == ref.shape
    assert rel_err(got.detach().cpu(), ref.detach()) < TOL
    (got * proj.float().to(DEV)).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) < TOL
    assert rel_err(wd.grad.cpu(), wr.grad) < TOL
    assert rel_err(bd.grad.cpu(), br.grad) < TOL
    # no-grad fast path with the bias fused
    with torch.no_grad():
        assert rel_err(conv.conv2d(xd.detach(), wd.detach(), bd.detach()).cpu(), ref.detach()) < TOL


@pytest.mark.parametrize("co,ci,k", [(128, 64, 3), (512, 1024, 3), (192, 256, 1), (64, 64, 1)])
def test_device_packer_equals_the_tensor_op_packer(co, ci, k):
    w = torch.randn(co, ci, k, k, generator=torch.Generator().manual_seed(co + k)).to(DEV)
    assert torch.equal(conv.pack_stream(w).view(-1), conv.pack_stream_torch(w).reshape(-1))
    assert torch.equal(conv.pack_stream(w, transposed=True).view(-1), conv.pack_stream_torch(conv._transposed(w)).reshape(-1))


@pytest.mark.parametrize("M,ci,co", [(20000, 256, 256), (33333, 128, 768), (16400, 768, 128), (70000, 64, 256)])
def test_linear_on_the_convolution_kernel(M, ci, co):
    """ops.linear: forward and data gradient as a 1x1 convolution over the rows (h3d_conv_x3), weight + bias gradient on
    h3d_wgrad_x3_bias, against float64."""
    lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, ci, generator=g, dtype=torch.float64)
    w = torch.randn(co, ci, generator=g, dtype=torch.float64) / ci ** 0.5
    b = torch.randn(co, generator=g, dtype=torch.float64)
    p = torch.randn(M, co, generator=g, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    (F.linear(xr, wr, br) * p).sum().backward()
    xd, wd, bd = (t.float().to(DEV).requires_grad_(True) for t in (x, w, b))
    assert lin._native_ok(co, ci)
    y = lin.linear(xd, wd, bd)
    assert rel_err(y.detach().cpu(), F.linear(x, w, b)) < TOL
    (y * p.float().to(DEV)).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) < TOL and rel_err(wd.grad.cpu(), wr.grad) < TOL and rel_err(bd.grad.cpu(), br.grad) < TOL
    with torch.no_grad():
        assert rel_err(lin.linear(xd.detach(), wd.detach(), bd.detach()).cpu(), F.linear(x, w, b)) < TOL


def test_conv_takes_channel_slices_without_copying():
    """Backward of a skip concatenation hands the convolutions channel slices of a wider channels-last tensor: they are read
    in place through the row stride (no dense copy), with the same results."""
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(2, 192, 10, 6, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 128, 3, 3, generator=g) / 34.0).to(DEV)
    sl = wide[:, 64:]
    rows, ld = conv._rows(sl)
    assert ld == 192 and rows.data_ptr() == sl.data_ptr()
    ref = F.conv2d(sl.double().cpu(), w.double().cpu(), padding=1)
    assert rel_err(conv.conv2d(sl, w).cpu(), ref) < TOL
    gy = torch.randn(2, 128, 10, 6, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)[:, 32:96]
    xr = sl.double().cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    (F.conv2d(xr, wr, padding=1) * gy.double().cpu()).sum().backward()
    xd, wd = sl.detach().requires_grad_(True), w.clone().requires_grad_(True)
    (conv.conv2d(xd, wd) * gy).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) < TOL and rel_err(wd.grad.cpu(), wr.grad) < TOL


def test_conv_double_backward_of_an_r1_style_penalty():
    g = torch.Generator().manual_seed(7)
    B, H, W, c = 2, 12, 6, 64
    x = torch.randn(B, c, H, W, generator=g, dtype=torch.float64)
    w1 = torch.randn(128, c, 3, 3, generator=g, dtype=torch.float64) / (c * 9) ** 0.5
    w2 = torch.randn(64, 128, 3, 3, generator=g, dtype=torch.float64) / (128 * 9) ** 0.5
    w3 = torch.randn(64, 64, 1, 1, generator=g, dtype=torch.float64) / 8.0

    def net(cv, x, w1, w2, w3):
        h = F.leaky_relu(cv(x, w1), 0.2)
        h = cv(F.leaky_relu(cv(h, w2) + cv(x, w3), 0.2), w3)
        return h

    ref_in = [t.clone().requires_grad_(True) for t in (x, w1, w2, w3)]
    out = net(lambda a, b: F.conv2d(a, b, padding=b.shape[2] // 2), *ref_in)
    (gx,) = torch.autograd.grad(out.sum(), ref_in[0], create_graph=True)
    pen = gx.pow(2).sum() + out.pow(2).mean()
    ref_g = torch.autograd.grad(pen, ref_in[1:])
    dev_in = [t.float().to(DEV).requires_grad_(True) for t in (x, w1, w2, w3)]
    out_d = net(conv.conv2d, *dev_in)
    assert rel_err(out_d.detach().cpu(), out.detach()) < TOL
    (gxd,) = torch.autograd.grad(out_d.sum(), dev_in[0], create_graph=True)
    assert rel_err(gxd.detach().cpu(), gx.detach()) < TOL
    pen_d = gxd.pow(2).sum() + out_d.pow(2).mean()
    assert abs(float(pen_d.detach()) - float(pen.detach())) < TOL * abs(float(pen.detach()))
    got_g = torch.autograd.grad(pen_d, dev_in[1:])
    for a, b, name in zip(got_g, ref_g, ("w1", "w2", "w3")):
        assert rel_err(a.cpu(), b) < TOL, name


def test_discriminator_uses_the_native_convolutions_and_matches_the_library_path(monkeypatch):
    """UNetDiscriminator at a config-4-like geometry: the native path is taken for the 64-multiple convolutions, and outputs,
    weight gradients and the R1 double backward agree with the torch / MIOpen path on the same weights."""
    disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
    trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
    torch.manual_seed(5)
    D = disc.UNetDiscriminator(latent_dim=64, gen_height=64, gen_width=32, label_dim=5, discriminator_blocks=4).to(DEV)
    with torch.no_grad():                      # settle the spectral-norm vectors (a fresh module's u / v are random), ...
        for _ in range(5):
            D(torch.zeros(1, 3, 64, 32, device=DEV), None, 1.0)
    D.eval()                                   # ... then freeze them: both runs below must see the same weights
    g = torch.Generator().manual_seed(6)
    real = torch.randn(2, 3, 64, 32, generator=g).clamp(-1, 1).to(DEV)
    fake = torch.randn(2, 3, 64, 32, generator=g).clamp(-1, 1).to(DEV)
    gt = torch.randint(0, 5, (2, 64, 32), generator=g).to(DEV)
    meta = dict(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, label_dim=5)
    calls = []
    real_run = conv._run_conv
    monkeypatch.setattr(conv, "_run_conv", lambda *a, **k: (calls.append(1), real_run(*a, **k))[1])
    res, grads = {}, {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("H3D_DISC_CONV", mode)
        calls.clear()
        res[mode] = trainers.discriminator_step(D, torch.optim.SGD(D.parameters(), lr=0.0), real, fake, gt, meta, do_r1=True,
                                                r1_mode="per_sample")
        grads[mode] = {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}
        assert (len(calls) > 40) == (mode == "hip"), (mode, len(calls))
    for k in ("loss", "gan", "r1", "segmentation"):
        assert abs(float(res["hip"][k]) - float(res["torch"][k])) < TOL * (abs(float(res["torch"][k])) + 1e-6), k
    assert set(grads["hip"]) == set(grads["torch"])
    worst = max(rel_err(grads["hip"][n], grads["torch"][n]) for n in grads["torch"] if float(grads["torch"][n].abs().max()) > 1e-6)
    assert worst < 5e-3, worst        # both sides carry their own rounding (MIOpen fp32 Winograd vs split bf16)


def test_misaligned_views_fall_back_to_a_copy_instead_of_raising():
    """the native fast paths were gated on dtype and shape only.  A contiguous view at a storage offset that is not
    a multiple of 16 bytes (a slice of a bias vector, rows of a larger buffer) must be copied, not rejected by the kernel."""
    import importlib
    lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
    rs = importlib.import_module("3dhumangan_amd.lib.components.resample")
    torch.manual_seed(0)
    big = torch.randn(4096 * 64 + 1, device="cuda")
    x = big[1:].view(4096, 64)                              # contiguous, first element 4 bytes past a 16-byte boundary
    w = torch.randn(128, 64, device="cuda")
    bias = torch.randn(129, device="cuda")[1:]
    assert x.data_ptr() % 16 and bias.data_ptr() % 16
    with torch.no_grad():
        y = lin.linear(x, w, bias)
    ref = torch.nn.functional.linear(x.double(), w.double(), bias.double())
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-3
    big2 = torch.randn(2 * 12 * 8 + 1, device="cuda")
    t = big2[1:].view(1, 12, 16)
    up = rs.bilinear_resize_cl(t, (4, 3), (16, 12))
    want = torch.nn.functional.interpolate(t.view(1, 4, 3, 16).permute(0, 3, 1, 2), (16, 12), mode="bilinear", align_corners=False)
    assert float((up.view(1, 16, 12, 16).permute(0, 3, 1, 2) - want).abs().max()) < 1e-5


@pytest.mark.parametrize("taps,slices,co,ci,bias", [(9, 170, 128, 128, True), (9, 2, 512, 512, False), (1, 37, 256, 256, True),
                                                     (9, 5, 64, 4, True), (1, 1, 8, 12, True), (4, 3, 260, 68, False)])
def test_the_slices_sum_kernel_equals_the_tensor_operations(taps, slices, co, ci, bias):
    """h3d_wgrad_reduce: the slices' sum, the [k, k, Co, Ci] -> [Co, Ci, k, k] layout change and the bias gradient's sum in one launch."""
    g = torch.Generator().manual_seed(taps * 1000 + slices)
    partial = torch.randn(taps, slices, co, ci, generator=g).to(DEV)
    colsum = torch.randn(slices, co, generator=g).to(DEV) if bias else None
    out = conv.reduce_slices(partial, colsum, taps, slices, co, ci, (co, ci, taps))
    dw, db = out if bias else (out, None)
    want = partial.double().sum(dim=1).permute(1, 2, 0)
    assert dw.shape == (co, ci, taps) and dw.is_contiguous()
    assert rel_err(dw, want) < 1e-6
    if bias:
        assert rel_err(db, colsum.double().sum(dim=0)) < 1e-6
    again = conv.reduce_slices(partial, colsum, taps, slices, co, ci, (co, ci, taps))
    assert torch.equal(dw, again[0] if bias else again)          # deterministic


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("layout", ["nchw", "channels_last", "slice", "one_channel", "column", "row"])
def test_channel_padding_kernel(layout, dtype):
    """h3d_pad_channels_cl: zero-padded channels in channels-last from any input layout, and the closed _PadChannels /
    _NarrowChannels pair through a double backward."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 12, 20
    if layout == "nchw":
        x = torch.randn(B, 3, H, W, generator=g)
    elif layout == "channels_last":
        x = torch.randn(B, 26, H, W, generator=g).contiguous(memory_format=torch.channels_last)
    elif layout == "slice":
        x = torch.randn(B, 64, H, W, generator=g).contiguous(memory_format=torch.channels_last)[:, :26]
    elif layout == "column":            # [B, P, C] rows as a [B, C, P, 1] image: the size-one dimension's stride says nothing
        x = torch.randn(B, 11, 32, generator=g).permute(0, 2, 1).unsqueeze(-1)
    elif layout == "row":
        x = torch.randn(B, 11, 32, generator=g).permute(0, 2, 1).unsqueeze(2)
    else:
        x = torch.randn(B, 1, H, W, generator=g)
    x = x.to(DEV).to(dtype)
    H, W = x.shape[2:]
    out = conv._pad_channels(x, 64)
    assert out.shape == (B, 64, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    C = x.shape[1]
    assert torch.equal(out[:, :C], x) and float(out[:, C:].abs().max()) == 0.0
    if dtype == torch.float32:
        xr = x.clone().requires_grad_(True)
        y = conv._PadChannels.apply(xr, 64)
        w = torch.randn(B, 64, H, W, generator=g).to(DEV)
        (gx,) = torch.autograd.grad((y * w).sum(), xr, create_graph=True)          # = w[:, :C], through _NarrowChannels
        assert torch.equal(gx, w[:, :C])
        v = torch.randn(B, 64, H, W, generator=g).to(DEV).requires_grad_(True)
        z = conv._NarrowChannels.apply(v, C)
        (gv,) = torch.autograd.grad((z * xr).sum(), v, create_graph=True)          # = pad(xr): depends on xr
        assert gv.is_contiguous(memory_format=torch.channels_last) and torch.equal(gv[:, :C], xr) and float(gv.detach()[:, C:].abs().max()) == 0.0
        (gxx,) = torch.autograd.grad((gv * w).sum(), xr)                           # second order: back through the pair
        assert torch.equal(gxx, w[:, :C])


@pytest.mark.parametrize("B,H,W,ci,co,k,half", [(1, 16, 8, 512, 512, 3, False), (2, 8, 4, 256, 512, 3, True), (1, 40, 33, 128, 256, 1, False),
                                                 (1, 64, 32, 256, 256, 1, True)])
def test_blocking_for_the_pixel_count_changes_no_bit(B, H, W, ci, co, k, half):
    """h3d_conv_x3_nt_for picks the tiles per output block that still fill the chip (the discriminator's low-resolution layers);
    h3d_conv_x3_pack_nt / h3d_conv_x3_ex at NT = 2, 4, 8 return the SAME bits (the blocking reorders no sum), and _run_conv -- which
    asks the helper -- matches float64."""
    lib = importlib.import_module("3dhumangan_amd._lib")
    L = lib.load()
    g = torch.Generator().manual_seed(ci + co + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    if half:
        x = x.half()
    outs = []
    for nt in (8, 4, 2):
        stream = torch.empty((1 if half else 2) * w.numel(), device=DEV, dtype=torch.int16)
        lib.check(L.h3d_conv_x3_pack_nt(lib.ptr(w), lib.ptr(stream), co, ci, k, 0, 2 if half else 0, nt, lib.stream_handle()), "pack")
        out = torch.empty((B, co, H, W), device=DEV, dtype=x.dtype, memory_format=torch.channels_last)
        lib.check(L.h3d_conv_x3_ex(2 if half else 0, lib.ptr(x), lib.ptr(stream), lib.ptr(b), None, lib.ptr(out), None, None, 1, B, H, W,
                                   ci, co, k, ci, co, 0, nt, lib.stream_handle()), "conv")
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    nt = L.h3d_conv_x3_nt_for(ci, co, B * H * W)
    tiles = (B * H * W + 127) // 128
    assert nt in (2, 4, 8) and (nt == 2 or tiles * (co // (32 * nt)) >= 256)
    conv_mod_planes = conv.AMP_WEIGHT_PLANES
    got = conv._run_conv(x, w, b)
    if (not half or conv_mod_planes == 1) and L.h3d_conv_x3_slices(ci, co, k, B * H * W, nt) == 1:      # K-slices add in another order
        assert torch.equal(got, outs[0])
    wq = w.half().double() if half else w.double()
    assert rel_err(got.double().cpu(), F.conv2d(x.double(), wq, b.double(), padding=k // 2).cpu()) < (2e-3 if half else TOL)
    # a blocking the channel count does not divide is refused
    assert L.h3d_conv_x3_pack_nt(lib.ptr(w), lib.ptr(stream), 64, 64, 1, 0, 0, 4, lib.stream_handle()) != 0


def test_runs_are_bit_identical_with_several_workgroups_per_cu():
    """The weight ring's write-after-read safety in conv_x3.hip is by construction (lgkmcnt(0) before the stage barrier): with the
    distance argument of the engines, 1-4 % of these launches (narrow blocking -> four workgroups per CU, the moments epilogue of
    one loading the LDS pipe under another's k-loop) returned a tile computed from a half-refilled stage."""
    lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
    g = torch.Generator().manual_seed(3)
    for (M, Co, Ci, dt) in ((524288, 64, 256, torch.float16), (524288, 256, 256, torch.float32)):
        x = torch.randn(M, Ci, generator=g).to(DEV, dt)
        w, b = (torch.randn(Co, Ci, generator=g) * 0.06).to(DEV), torch.randn(Co, generator=g).to(DEV)
        r = torch.randn(M, Co, generator=g).to(DEV, dt)
        y0, p0 = lin.gemm_x3(x, w, b, add=r, moments=True)
        bad = 0
        for _ in range(120):
            y, p = lin.gemm_x3(x, w, b, add=r, moments=True)
            bad += int(not (torch.equal(y, y0) and torch.equal(p, p0)))
        assert bad == 0, (M, Co, Ci, dt, bad)


@pytest.mark.parametrize("B,H,W,ci,co,k,half,slices", [(4, 16, 8, 512, 512, 3, False, 8), (4, 8, 4, 512, 512, 3, True, 9), (1, 13, 7, 256, 128, 3, False, 3),
                                                        (2, 32, 16, 1024, 256, 1, False, 4), (4, 32, 16, 256, 512, 3, True, 2), (1, 5, 5, 64, 64, 3, False, 16)])
def test_k_slices(B, H, W, ci, co, k, half, slices):
    """h3d_conv_x3_ex with the tap x chunk loop cut into K-slices (own workgroups, fp32 partial sums, a second launch adding them in
    slice order with bias and addend): against float64 and against the unsliced launch, fp32 and f16, with more slices asked for than
    the loop has pieces to give; h3d_conv_x3_slices asks for them only where the grid leaves CUs idle."""
    lib = importlib.import_module("3dhumangan_amd._lib")
    L = lib.load()
    g = torch.Generator().manual_seed(ci + co + H + slices)
    dt = torch.float16 if half else torch.float32
    x = torch.randn(B, ci, H, W, generator=g).to(DEV, dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    r = torch.randn(B, co, H, W, generator=g).to(DEV, dt).contiguous(memory_format=torch.channels_last)
    nt = L.h3d_conv_x3_nt_for(ci, co, B * H * W)
    stream = torch.empty((1 if half else 2) * w.numel(), device=DEV, dtype=torch.int16)
    lib.check(L.h3d_conv_x3_pack_nt(lib.ptr(w), lib.ptr(stream), co, ci, k, 0, 2 if half else 0, nt, lib.stream_handle()), "pack")
    outs = []
    for s_ in (1, slices):
        out = torch.full((B, co, H, W), float("nan"), device=DEV, dtype=dt).contiguous(memory_format=torch.channels_last)
        work = torch.full((max(s_, 1), B * H * W, co), float("nan"), device=DEV) if s_ > 1 else None
        lib.check(L.h3d_conv_x3_ex(2 if half else 0, lib.ptr(x), lib.ptr(stream), lib.ptr(b), lib.ptr(r), lib.ptr(out), None, lib.ptr(work), s_,
                                   B, H, W, ci, co, k, ci, co, co, nt, lib.stream_handle()), "conv")
        outs.append(out)
    wq = w.half().double() if half else w.double()
    want = F.conv2d(x.double(), wq, b.double(), padding=k // 2) + r.double()
    tol = 2e-3 if half else TOL
    assert rel_err(outs[0].double().cpu(), want.cpu()) < tol and rel_err(outs[1].double().cpu(), want.cpu()) < tol
    assert rel_err(outs[1].double(), outs[0].double()) < (1e-3 if half else 1e-5)
    auto = L.h3d_conv_x3_slices(ci, co, k, B * H * W, nt)
    assert auto >= 1 and (auto == 1 or ((B * H * W + 127) // 128) * (co // (32 * nt)) < 160)
    # slices without a workspace, or together with moments, are refused
    assert L.h3d_conv_x3_ex(0, lib.ptr(x), lib.ptr(stream), None, None, lib.ptr(outs[0]), None, None, 4, B, H, W, ci, co, k, ci, co, 0, nt,
                            lib.stream_handle()) != 0
