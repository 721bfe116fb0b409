"""GPU parity of the modulated-convolution plugin ops (P3) against golden vectors from the reference / the oracle."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
m3 = importlib.import_module("3dhumangan_amd.lib.components.map3d_layers")
cips = importlib.import_module("3dhumangan_amd.lib.components.cips_layers")
DEV = "cuda"


def test_modconv1x1_golden():
    g = load_golden("plugin_ops")["modconv1x1"]
    layer = m3.SpatialStyleModLayer(in_channel=24, out_channel=40, style_dim=16)
    layer.load_state_dict(g["state"])
    layer = layer.to(DEV)
    out = layer(g["x"].to(DEV), g["style"].to(DEV))
    assert rel_err(out.cpu(), g["out"]) < 1e-5


@pytest.mark.parametrize("ks", [1, 3])
def test_modconv2d_golden(ks):
    g = load_golden("plugin_ops")[f"modconv2d_k{ks}"]
    layer = cips.StyleModLayer(in_channel=12, out_channel=20, kernel_size=ks, style_dim=10)
    layer.load_state_dict(g["state"])
    layer = layer.to(DEV)
    out = layer(g["x"].to(DEV), g["style"].to(DEV))
    assert out.shape == g["out"].shape
    assert rel_err(out.cpu(), g["out"]) < 1e-5


@pytest.mark.parametrize("cin,cout,s,rows,demod", [(256, 256, 256, 300, True), (64, 200, 48, 65, True), (33, 7, 5, 1, False)])
def test_modconv1x1_vs_oracle(cin, cout, s, rows, demod):
    torch.manual_seed(cin + rows)
    layer = m3.SpatialStyleModLayer(in_channel=cin, out_channel=cout, style_dim=s, demodulate=demod)
    with torch.no_grad():
        layer.bias.add_(0.1 * torch.randn_like(layer.bias))
    x, st = torch.randn(2, rows, cin), torch.randn(2, rows, s)
    ref = O.modconv1x1_pixelwise(x.double(), st.double(), layer.weight[0, 0].detach().double(), layer.bias[0, 0].detach().double(),
                                 layer.affine.weight.detach().double(), layer.affine.bias.detach().double(), demodulate=demod)
    out = layer.to(DEV)(x.to(DEV), st.to(DEV))
    assert rel_err(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("cin,cout,k,hw", [(64, 64, 3, (17, 13)), (256, 128, 1, (8, 8)), (40, 300, 3, (9, 70)), (8, 8, 5, (6, 6))])
def test_modconv2d_vs_oracle(cin, cout, k, hw):
    torch.manual_seed(cin + k)
    layer = cips.StyleModLayer(in_channel=cin, out_channel=cout, kernel_size=k, style_dim=16)
    with torch.no_grad():
        layer.bias.add_(0.1 * torch.randn_like(layer.bias))
    x, st = torch.randn(2, cin, *hw), torch.randn(2, 16)
    ref = O.modconv2d_grouped(x.double(), st.double(), layer.weight.detach().double(), layer.bias.detach().double(),
                              layer.geo_feature.weight.detach().double(), layer.geo_feature.bias.detach().double())
    out = layer.to(DEV)(x.to(DEV), st.to(DEV))
    assert rel_err(out.cpu(), ref) < 2e-5
    # 2-D / 3-D inputs of the reference API (k = 1 semantics)
    if k == 1:
        x2 = torch.randn(2, cin)
        ref2 = O.modconv2d_grouped(x2[:, :, None, None].double(), st.double(), layer.weight.detach().cpu().double(),
                                   layer.bias.detach().cpu().double(), layer.geo_feature.weight.detach().cpu().double(),
                                   layer.geo_feature.bias.detach().cpu().double())[:, :, 0, 0]
        assert rel_err(layer(x2.to(DEV), st.to(DEV)).cpu(), ref2) < 2e-5


def _grads(out, cot, tensors):
    return torch.autograd.grad(out, tensors, cot)


@pytest.mark.parametrize("cin,cout,s,rows,demod", [(64, 96, 48, 65, True), (33, 7, 5, 3, False)])
def test_modconv1x1_gradients_vs_oracle(cin, cout, s, rows, demod):
    """Every input and parameter gets the gradient of the oracle's float64 restatement.  With gradients the layer is composed of the
    package's dense-layer primitives (split-bf16 matrix-core GEMMs where the problem is large enough, ~2e-5 per product), so
    the bounds are those of that arithmetic -- still 10x inside the 1e-3 budget; the no-grad path is the fused fp32 kernel."""
    torch.manual_seed(cin)
    layer = m3.SpatialStyleModLayer(in_channel=cin, out_channel=cout, style_dim=s, demodulate=demod)
    x, st = torch.randn(2, rows, cin), torch.randn(2, rows, s)
    cot = torch.randn(2, rows, cout)
    names = ["weight", "bias", "affine.weight", "affine.bias"]
    ps = dict(layer.named_parameters())
    ref_leaves = [t.detach().double().requires_grad_() for t in (x, st, *[ps[n] for n in names])]
    rx, rst, rw, rb, raw, rab = ref_leaves
    ref = O.modconv1x1_pixelwise(rx, rst, rw[0, 0], rb[0, 0], raw, rab, demodulate=demod)
    ref_g = _grads(ref, cot.double(), ref_leaves)
    layer = layer.to(DEV)
    ps = dict(layer.named_parameters())
    xd, sd = x.to(DEV).requires_grad_(), st.to(DEV).requires_grad_()
    out = layer(xd, sd)
    assert out.requires_grad and rel_err(out.detach().cpu(), ref.detach()) < 1e-4
    got = _grads(out, cot.to(DEV), [xd, sd] + [ps[n] for n in names])
    for name, a, b in zip(["x", "style"] + names, got, ref_g):
        assert rel_err(a.cpu(), b) < 3e-4, name
    # only some inputs need a gradient / none does
    out = layer(x.to(DEV), sd)
    (g_s,) = _grads(out, cot.to(DEV), [sd])
    assert rel_err(g_s.cpu(), ref_g[1]) < 3e-4
    with torch.no_grad():
        assert not layer(xd, sd).requires_grad


@pytest.mark.parametrize("cin,cout,k,hw,dim", [(64, 64, 3, (9, 7), 4), (40, 72, 1, (5, 6), 4), (32, 48, 1, None, 2), (32, 48, 1, 11, 3)])
def test_modconv2d_gradients_vs_oracle(cin, cout, k, hw, dim):
    torch.manual_seed(cin + k + dim)
    layer = cips.StyleModLayer(in_channel=cin, out_channel=cout, kernel_size=k, style_dim=16)
    x = torch.randn(2, cin, *hw) if dim == 4 else torch.randn(2, cin) if dim == 2 else torch.randn(2, hw, cin)
    st = torch.randn(2, 16)
    names = ["weight", "bias", "geo_feature.weight", "geo_feature.bias"]
    ps = dict(layer.named_parameters())
    ref_leaves = [t.detach().double().requires_grad_() for t in (x, st, *[ps[n] for n in names])]
    rx, rst, rw, rb, rgw, rgb = ref_leaves
    img = rx if dim == 4 else rx[:, :, None, None] if dim == 2 else rx.permute(0, 2, 1).unsqueeze(-1)
    ref = O.modconv2d_grouped(img, rst, rw, rb, rgw, rgb)
    ref = ref if dim == 4 else ref[:, :, 0, 0] if dim == 2 else ref[:, :, :, 0].permute(0, 2, 1)
    cot = torch.randn(ref.shape)
    ref_g = _grads(ref, cot.double(), ref_leaves)
    layer = layer.to(DEV)
    ps = dict(layer.named_parameters())
    xd, sd = x.to(DEV).requires_grad_(), st.to(DEV).requires_grad_()
    out = layer(xd, sd)
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-4        # with gradients: native split-bf16 convolution (~2e-5 per product)
    with torch.no_grad():
        assert rel_err(layer(xd, sd).cpu(), ref.detach()) < 2e-5   # without: the fused fp32 kernel
    got = _grads(out, cot.to(DEV), [xd, sd] + [ps[n] for n in names])
    for name, a, b in zip(["x", "style"] + names, got, ref_g):
        assert rel_err(a.cpu(), b) < 3e-4, name
    # second order through the same primitives (they are closed under differentiation): d/dx of a gradient norm
    if dim == 4:
        out2 = layer(xd, sd)
        (gx,) = torch.autograd.grad(out2, [xd], cot.to(DEV), create_graph=True)
        (hx,) = torch.autograd.grad(gx.square().sum(), [ps["weight"]])
        rout = O.modconv2d_grouped(img, rst, rw, rb, rgw, rgb)
        (rgx,) = torch.autograd.grad(rout, [rx], cot.double(), create_graph=True)
        (rhx,) = torch.autograd.grad(rgx.square().sum(), [rw])
        assert rel_err(hx.cpu(), rhx) < 1e-3
