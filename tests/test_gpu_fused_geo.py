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


def test_pruned_search_on_the_sorted_mesh_equals_the_full_scan_at_render_scale(monkeypatch):
    """Round 4: h3d_mesh_sort + h3d_nearest_vertex_sorted (chunks of the Morton-sorted mesh skipped by bounding sphere) against
    the full scan of the unsorted mesh on the renderer's own points (4 x 96 x 96 rays x 64 jittered samples: 2.4 M points per
    pose, where exact distance ties DO occur), a mirror-symmetric mesh with duplicated vertices (ties by construction), and the
    oracle's brute force on a subset.  Indices are integers: torch.equal."""
    vr = importlib.import_module("3dhumangan_amd.lib.generators.volume_rendering")
    cond = synthetic.make_conditions(2, n_vertices=6890, seed=11)
    v = cond["vertices"]
    v[1, :3000, 0] = v[1, :3000, 0].abs() + 0.01                    # pose 1: vertices 3000..5999 mirror 0..2999 in x = 0,
    v[1, 3000:6000] = v[1, :3000] * torch.tensor([-1.0, 1.0, 1.0])  # 6000.. duplicate some of them
    v[1, 6000:] = v[1, 100:990]
    jit = torch.rand(2, 96 * 96, 64, 1, generator=torch.Generator().manual_seed(3))
    c = dev_dict(cond)
    pts, _ = vr.sample_rays(c["intrinsics"][:, 0, 0], c["scales"], c["cam2world_matrices"], 64, (96, 96), -0.5, 0.55, jitter=jit.to(DEV))
    pts = pts.reshape(2, -1, 3).clone()
    pts[1, ::7, 0] = 0.0                                            # on the symmetry plane: every nearest vertex is tied
    monkeypatch.setattr(smpl, "PRUNE", True)
    pruned = smpl.nearest_vertex(pts, c["vertices"])
    monkeypatch.setattr(smpl, "PRUNE", False)
    full = smpl.nearest_vertex(pts, c["vertices"])
    assert torch.equal(pruned, full)
    # round 5: the same search with compact wave tiles (8 x 8 rays x 4 samples: h3d_nearest_vertex_sorted_rays) -- the same indices
    monkeypatch.setattr(smpl, "PRUNE", True)
    tiled = smpl.nearest_vertex(pts, c["vertices"], ray_shape=(96, 96, 64))
    assert torch.equal(tiled, full)
    monkeypatch.setattr(smpl, "TILED", False)
    assert torch.equal(smpl.nearest_vertex(pts, c["vertices"], ray_shape=(96, 96, 64)), full)
    monkeypatch.setattr(smpl, "TILED", True)
    for hr, wr, ss in ((24, 16, 8), (20, 16, 8), (16, 12, 8), (16, 16, 6), (8, 8, 4)):     # dividing and non-dividing grids
        n = hr * wr * ss
        q = pts[:, 5000: 5000 + n].contiguous()
        monkeypatch.setattr(smpl, "PRUNE", False)
        want = smpl.nearest_vertex(q, c["vertices"])
        monkeypatch.setattr(smpl, "PRUNE", True)
        assert torch.equal(smpl.nearest_vertex(q, c["vertices"], ray_shape=(hr, wr, ss)), want), (hr, wr, ss)
    monkeypatch.setattr(smpl, "PRUNE", False)
    sub = torch.randperm(pts.shape[1], generator=torch.Generator().manual_seed(1))[:6000]
    sub = torch.cat([sub, torch.arange(0, 7 * 300, 7)])             # some of the tied ones too
    _, ridx = O.nearest_vertex(pts[:, sub].cpu().float(), cond["vertices"].float())
    assert torch.equal(pruned[:, sub].cpu().long(), ridx)
    # the feature kernel on the sorted mesh: same index, same features as on the unsorted one
    monkeypatch.setattr(smpl, "PRUNE", True)
    args = (pts[:, :70000], c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"], c["lbs_weights"], False)
    g1, i1 = smpl.get_geo_features(*args, return_index=True)
    monkeypatch.setattr(smpl, "PRUNE", False)
    g2, i2 = smpl.get_geo_features(*args, return_index=True)
    assert torch.equal(i1, i2) and torch.equal(i1, pruned[:, :70000]) and torch.equal(g1, g2)


@pytest.mark.parametrize("V", [1, 63, 64, 65, 700, 6890, 10000])
def test_mesh_sort_is_a_permutation_with_valid_spheres(V):
    """h3d_mesh_sort: the workspace holds every vertex exactly once with its original index, padding at 3e18, and every chunk's
    sphere contains its vertices."""
    g = torch.Generator().manual_seed(V)
    verts = torch.randn(3, V, 3, generator=g).to(DEV)
    verts[2] = verts[2, :1]                                           # a degenerate pose: all vertices identical
    ws = smpl.sort_mesh(verts)
    Vpad = (V + 63) // 64 * 64
    ws = ws.view(3, Vpad + Vpad // 64, 4)
    ids = ws[:, :V, 3].contiguous().view(torch.int32).long()
    assert torch.equal(ids.sort(dim=1).values, torch.arange(V, device=DEV).expand(3, V))
    assert torch.equal(ws[:, :V, :3], torch.gather(verts, 1, ids[..., None].expand(3, V, 3)))
    assert bool((ws[:, V:Vpad, :3] > 1e18).all())
    for c in range(Vpad // 64):
        vv = ws[:, c * 64:min((c + 1) * 64, V), :3]
        cen, rad = ws[:, Vpad + c, :3], ws[:, Vpad + c, 3]
        assert bool(((vv - cen[:, None]).double().norm(dim=-1) <= rad[:, None].double()).all())
