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
