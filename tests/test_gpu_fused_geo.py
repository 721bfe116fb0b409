"""A4 inside the fused render (SURVEY 8a rows A4 + A5 + A6; north_star: "the ray-sample / MLP / alpha-composite loop is a fused
kernel"): h3d_nearest_vertex writes only the K = 1 nearest-vertex index, h3d_render_fused_x2_geo / _x3_geo build the 31
geometry features (lib/components/smpl.py:210-249) in the field kernel's prologue.  Checked against the oracle and against the
two-kernel path (h3d_geo_features + h3d_render_fused_*)."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import rel_err
from test_gpu_field import random_state

pytestmark = pytest.mark.gpu
smpl = importlib.import_module("3dhumangan_amd.lib.components.smpl")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")
DEV = "cuda"
TOL = 1e-3


def dev_dict(d):
    return {k: v.to(DEV) for k, v in d.items()}


@pytest.mark.parametrize("V,N", [(6890, 5000), (777, 1300), (64, 257), (1, 40)])
def test_nearest_vertex_is_the_oracles_index(V, N):
    """Integer work is bit-exact: the search-only kernel returns the oracle's arg-min (first index on ties) and the very index
    h3d_geo_features reports."""
    cond = synthetic.make_conditions(2, n_vertices=V, seed=V)
    g = torch.Generator().manual_seed(N)
    pts = torch.rand(2, N, 3, generator=g) * 2.4 - 1.2
    if V >= 8:
        pts[:, :8] = cond["vertices"][:, :8]                        # exact hits
        pts[:, 8] = 0.5 * (cond["vertices"][:, 0] + cond["vertices"][:, 1])      # an (almost) tie
    _, ridx = O.nearest_vertex(pts.float(), cond["vertices"].float())
    c = dev_dict(cond)
    idx = smpl.nearest_vertex(pts.to(DEV), c["vertices"])
    assert idx.dtype == torch.int32 and torch.equal(idx.cpu().long(), ridx)
    _, idx2 = smpl.get_geo_features(pts.to(DEV), c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"],
                                    c["lbs_weights"], False, return_index=True)
    assert torch.equal(idx, idx2)


@pytest.mark.parametrize("engine", ["f16x2", "f16x3"])
@pytest.mark.parametrize("S,R,hidden,legacy", [(8, 20, 32, False), (16, 30, 64, True), (32, 9, 64, False), (64, 5, 256, False),
                                               (64, 7, 256, True), (128, 3, 128, False), (96, 3, 200, True)])
def test_render_geo_vs_oracle_and_vs_the_two_kernel_path(S, R, hidden, legacy, engine):
    state, net = random_state(hidden, hidden, seed=S + hidden, precision=engine)
    with torch.no_grad():
        net.sigma_layer.weight.mul_(40.0)
        state["neural_field.sigma_layer.weight"] = net.sigma_layer.weight.detach().cpu().clone()
    B, N, V = 2, R * S, 500
    cond = synthetic.make_conditions(B, n_vertices=V, seed=R)
    g = torch.Generator().manual_seed(R)
    pts = torch.rand(B, N, 3, generator=g) * 2 - 1
    freq = torch.randn(B, 4 * hidden, generator=g) * 0.5
    phase = torch.randn(B, 4 * hidden, generator=g)
    z = torch.sort(torch.rand(B, R, S, 1, generator=g) + 11, dim=2).values
    noise = torch.randn(B, R, S, 1, generator=g) * 0.3
    dirs = torch.zeros(B, N, 3)
    dirs[..., 2] = -1
    geo_ref = O.geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"], cond["fk_matrices"],
                             cond["lbs_weights"], legacy)
    sd = {k: v.double() for k, v in state.items()}
    field = O.neural_field(sd, pts.double(), freq.double(), phase.double(), geo_ref.double(), dirs.double(), 0.7)
    ref = O.ray_integration(field.reshape(B, R, S, -1), z.double(), noise.double(), "relu", True, False)
    c = dev_dict(cond)
    assert net.render_geo_supported(S)
    vik = smpl.vertex_inverse_transforms(c["fk_matrices"], c["lbs_weights"])
    idx = smpl.nearest_vertex(pts.to(DEV), c["vertices"])
    got = net.render_geo(pts.to(DEV), freq.to(DEV), phase.to(DEV), idx, c["skeletons_xyz"], c["vertices"], c["tpose_vertices"],
                         vik, None, z.to(DEV), S, legacy_mode=legacy, input_scaler=0.7, noise=noise.to(DEV), clamp_mode="relu",
                         last_back=True, white_back=False)
    geo = smpl.get_geo_features(pts.to(DEV), c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"],
                                c["lbs_weights"], legacy)
    two = net.render(pts.to(DEV), freq.to(DEV), phase.to(DEV), geo, None, z.to(DEV), S, input_scaler=0.7,
                     noise=noise.to(DEV), clamp_mode="relu", last_back=True, white_back=False)
    for a, b, t, nm in zip(got, ref, two, ("feats", "depth", "weights")):
        assert a.shape == b.shape, nm
        assert rel_err(a.cpu(), b) < TOL, nm
        # same arithmetic behind the features except their last bit (v_sqrt / reciprocal constants): far inside the budget
        assert rel_err(a.cpu(), t.cpu()) < 1e-4, nm


def test_generator_uses_the_fused_geometry_path_by_default():
    """Map3DGenerator.forward in eval mode: nearest-vertex search + render_geo (stage timer shows no feature tensor pass), same
    image as the two-kernel path and as the oracle."""
    from conftest import load_golden
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
    g = load_golden("gen_tiny_isolated_legacy")
    cfg = dict(g["meta"])
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    G = gens.Map3DGenerator(**cfg)
    G.load_state_dict(g["state"], strict=True)
    G = G.to(DEV).eval()
    G.set_device(DEV)
    assert G.fuse_geo and G.neural_field.render_geo_supported(cfg["num_steps"])
    cond = dev_dict(g["cond"])
    kw = dict(jitter=g["jitter"].to(DEV), noise=g["noise"].to(DEV))
    out = G.forward(g["z"].to(DEV), cond, **kw, **cfg)
    G.fuse_geo = False
    two = G.forward(g["z"].to(DEV), cond, **kw, **cfg)
    for k in ("rgbs", "rgbs_render"):
        assert rel_err(out[k].cpu(), g["out"][k]) < TOL
        assert rel_err(out[k].cpu(), two[k].cpu()) < 1e-4
