"""The device-side field packers (h3d_field_pack_x3_device / _x2_device, csrc/field_x3.hip: field_pack_kernel) against the host
packers: the same blob, bit for bit -- and the train-mode forward that nothing records (the D step's generator forward,
reference lib/trainers/phase_trainer.py:355-362) on the fused render they make possible."""
import importlib

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
DEV = "cuda"


def field(hidden, feature, seed, scale=1.0):
    torch.manual_seed(seed)
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=hidden, hidden_dim=hidden, geo_feature_dim=31, output_dim=feature + 4,
                                feature_dim=feature, num_blocks=4)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
            p.mul_(scale)
    return net.to(DEV).eval()


@pytest.mark.parametrize("precision", ["f16x3", "f16x2"])
@pytest.mark.parametrize("hidden,feature,scale", [(256, 32, 1.0), (128, 32, 1.0), (64, 64, 1.0), (40, 40, 1.0), (200, 24, 37.5),
                                                  (256, 32, 2.0 ** -20), (96, 160, 1.0)])
def test_device_blob_is_the_host_blob(hidden, feature, scale, precision):
    net = field(hidden, feature, seed=hidden + feature, scale=scale)
    net.precision = precision
    net.device_pack = False
    host = net.packed_weights(DEV).clone()
    net._packed.clear()
    net.device_pack = True
    dev = net.packed_weights(DEV)
    torch.cuda.synchronize()
    assert dev.is_cuda and dev.shape == host.shape
    a, b = host.view(torch.int32).cpu(), dev.view(torch.int32).cpu()
    assert torch.equal(a, b), f"{int((a != b).sum())} of {a.numel()} words differ, first at {int((a != b).nonzero()[0])}"


def test_an_all_zero_matrix_and_a_zero_row_pack_alike():
    net = field(64, 32, seed=3)
    with torch.no_grad():
        net.network[1].layer.weight.zero_()
        net.color_layer_sine.layer.weight[5].zero_()
        net.sigma_layer.weight.zero_()
    for precision in ("f16x3", "f16x2"):
        net.precision = precision
        net._packed.clear()
        net.device_pack = False
        host = net.packed_weights(DEV).clone()
        net._packed.clear()
        net.device_pack = True
        assert torch.equal(host.view(torch.int32), net.packed_weights(DEV).view(torch.int32))


def test_the_blob_follows_the_weights_without_a_host_copy():
    """A parameter update (version bump) repacks on the device; the render sees the new weights."""
    net = field(64, 32, seed=5)
    net.precision = "f16x3"
    first = net.packed_weights(DEV).clone()
    assert net.packed_weights(DEV).data_ptr() == net._packed["h3d_field_pack_x3"][1].data_ptr()      # cached while nothing changes
    with torch.no_grad():
        net.network[0].layer.weight.mul_(1.5)
    second = net.packed_weights(DEV)
    assert not torch.equal(first.view(torch.int32), second.view(torch.int32))
    net.device_pack = False
    net._packed.clear()
    assert torch.equal(net.packed_weights(DEV).view(torch.int32), second.view(torch.int32))


@pytest.mark.parametrize("tier,tol_render,tol_rgb", [("x3", 1e-4, 2e-4), ("x2", 1e-3, 1e-3)])
def test_unrecorded_train_forward_runs_the_fused_render(tier, tol_render, tol_rgb):
    """Train mode under no_grad (what the D step calls): the field + integration go through the fused render on device-packed
    weights -- same outputs as the reference module in train mode, same buffer updates -- and `H3D_TRAIN_FIELD=off` keeps
    lib/generators/differentiable.py."""
    from conftest import load_golden
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    g = load_golden("gen_train_mixed")
    cfg = dict(g["meta"])
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    outs = {}
    for mode in (tier, "off"):
        G = gens.Map3DGenerator(**cfg)
        G.load_state_dict(g["state"], strict=True)
        G = G.to(DEV)
        G.set_device(DEV)
        G.train()
        G.train_field = mode
        calls = []
        keep = G.neural_field.render_geo
        G.neural_field.render_geo = lambda *a, **k: (calls.append(G.neural_field.precision), keep(*a, **k))[1]
        cond = {k: v.to(DEV) for k, v in g["cond"].items()}
        idx = g["latent_indices"].to(DEV) if "latent_indices" in g else None
        with torch.no_grad():
            out = G(g["z"].to(DEV), cond, latent_indices=idx, jitter=g["jitter"].to(DEV), noise=g["noise"].to(DEV), **cfg)
        assert calls == ([] if mode == "off" else ["f16x3" if mode == "x3" else "f16x2"]), calls
        assert G.neural_field.precision == "f16x2"                      # the engine choice of the call does not stick
        assert rel_err(out["rgbs_render"].cpu(), g["out"]["rgbs_render"]) < tol_render
        assert rel_err(out["rgbs"].cpu(), g["out"]["rgbs"]) < tol_rgb
        sd = G.state_dict()
        for k, ref in g["buffers_after"].items():
            if ref.is_floating_point():
                assert rel_err(sd[k].cpu(), ref) < (1e-4 if mode != "x2" else 1e-3), k
        outs[mode] = out
        # with autograd recording the differentiable path runs, whatever the knob says
        calls.clear()
        z = g["z"].to(DEV).requires_grad_(True)
        rec = G(z, cond, latent_indices=idx, jitter=g["jitter"].to(DEV), noise=g["noise"].to(DEV), **cfg)
        assert calls == [] and rec["rgbs"].requires_grad
    assert rel_err(outs[tier]["rgbs"], outs["off"]["rgbs"]) < tol_rgb
