"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle / golden vectors."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

vr = importlib.import_module("3dhumangan_amd.lib.generators.volume_rendering")
smpl = importlib.import_module("3dhumangan_amd.lib.components.smpl")
resample = importlib.import_module("3dhumangan_amd.lib.components.resample")
bias_act_mod = importlib.import_module("3dhumangan_amd.lib.components.ops.bias_act")
upfirdn_mod = importlib.import_module("3dhumangan_amd.lib.components.ops.upfirdn2d")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")

DEV = "cuda"


def dev(t):
    return t.to(DEV)


def dev_dict(d):
    return {k: v.to(DEV) for k, v in d.items()}


# ------------------------------------------------------------------ A6

def test_ray_integration_golden():
    g = load_golden("ray_integration")
    for k, c in g.items():
        S, C, softplus, last_back, white_back = [int(v) for v in c["flags"]]
        f, d, w = vr.ray_integration(dev(c["field"]), dev(c["z"]), noise=dev(c["noise"]),
                                     clamp_mode="softplus" if softplus else "relu", last_back=bool(last_back),
                                     white_back=bool(white_back))
        assert rel_err(f.cpu(), c["feats"]) < 1e-5, k
        assert rel_err(d.cpu(), c["depth"]) < 1e-6, k
        assert rel_err(w.cpu(), c["weights"]) < 1e-5, k


@pytest.mark.parametrize("S,C", [(32, 387), (64, 259), (128, 259), (1, 3), (7, 6), (200, 1027)])
def test_ray_integration_vs_oracle(S, C):
    g = torch.Generator().manual_seed(S * 1000 + C)
    B, R = 2, 37
    field = torch.randn(B, R, S, C + 1, generator=g)
    field[..., -1] = field[..., -1] * 10 - 2
    z = torch.sort(torch.rand(B, R, S, 1, generator=g) + 11.0, dim=2).values
    for last_back in (False, True):
        ref = O.ray_integration(field.double(), z.double(), None, "relu", last_back, True)
        got = vr.ray_integration(dev(field), dev(z), noise_std=0, clamp_mode="relu", last_back=last_back,
                                 white_back=True)
        for a, b, name in zip(got, ref, ("feats", "depth", "weights")):
            assert rel_err(a.cpu(), b) < 2e-5, (name, last_back)


def test_ray_integration_properties_full_size():
    """cfg3-sized rays (S=64, F=256): weights are a partition of unity with last_back, and the op is linear in
    the feature channels."""
    g = torch.Generator().manual_seed(3)
    B, R, S, C = 1, 4608, 64, 259
    field = torch.randn(B, R, S, C + 1, generator=g).to(DEV)
    z = torch.sort(torch.rand(B, R, S, 1, generator=g) + 11.0, dim=2).values.to(DEV)
    f1, d1, w1 = vr.ray_integration(field, z, noise_std=0, clamp_mode="relu", last_back=True, white_back=False)
    assert float((w1.sum(2) - 1).abs().max()) < 1e-5
    assert float(w1.min()) >= 0
    assert float((d1 - (w1 * z).sum(2)).abs().max()) < 1e-4
    field2 = field.clone()
    field2[..., :-1] *= 2.0
    f2, _, w2 = vr.ray_integration(field2, z, noise_std=0, clamp_mode="relu", last_back=True, white_back=False)
    assert torch.equal(w1, w2)
    assert rel_err(f2.cpu(), (2 * f1).cpu()) < 1e-6
    ref = (w1 * field[..., :-1]).sum(2)
    assert rel_err(f1.cpu(), ref.cpu()) < 1e-5


def test_ray_integration_rejects_bad_mode():
    x = torch.zeros(1, 1, 4, 5, device=DEV)
    with pytest.raises(Exception):
        vr.ray_integration(x, torch.zeros(1, 1, 4, 1, device=DEV), clamp_mode=None)


# ------------------------------------------------------------------ A3

@pytest.mark.parametrize("name", ["gen_tiny_mixed", "gen_tiny_isolated_legacy"])
def test_ray_setup_golden(name):
    g = load_golden(name)
    cfg, cond = g["meta"], g["cond"]
    pts, zv = vr.sample_rays(dev(cond["intrinsics"][:, 0, 0]), dev(cond["scales"]), dev(cond["cam2world_matrices"]),
                             cfg["num_steps"], (cfg["render_width"], cfg["render_height"]), cfg["ray_start"],
                             cfg["ray_end"], jitter=dev(g["jitter"]))
    assert rel_err(pts.cpu(), g["stage"]["points"]) < 5e-6
    assert rel_err(zv.cpu(), g["stage"]["z_vals"]) < 1e-6


def test_ray_setup_full_size_vs_oracle():
    cond = synthetic.make_conditions(2, n_vertices=64, seed=1)
    jit = torch.rand(2, 96 * 48, 64, 1, generator=torch.Generator().manual_seed(0))
    ref_p, ref_z, _ = O.ray_setup(cond["intrinsics"][:, 0, 0], cond["scales"], cond["cam2world_matrices"], 96, 48, 64,
                                  -0.5, 0.55, jit)
    pts, zv = vr.sample_rays(dev(cond["intrinsics"][:, 0, 0]), dev(cond["scales"]), dev(cond["cam2world_matrices"]),
                             64, (48, 96), -0.5, 0.55, jitter=dev(jit))
    assert rel_err(pts.cpu(), ref_p) < 5e-6
    assert rel_err(zv.cpu(), ref_z) < 1e-6


# ------------------------------------------------------------------ A4

def _geo_check(points, cond, legacy):
    ref, ridx, rd2 = O.geo_features(points, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"],
                                    cond["fk_matrices"], cond["lbs_weights"], legacy, return_index=True)
    c = dev_dict(cond)
    got, idx = smpl.get_geo_features(dev(points), c["skeletons_xyz"], c["vertices"], c["tpose_vertices"],
                                     c["fk_matrices"], c["lbs_weights"], legacy, return_index=True)
    # integer work is bit-exact: same arithmetic order, first index wins
    assert torch.equal(idx.cpu().long(), ridx)
    assert rel_err(got.cpu(), ref) < 1e-5
    return got


def test_geo_features_golden_full_mesh():
    g = load_golden("geo_features")
    cond = synthetic.make_conditions(2, n_vertices=6890, seed=int(g["cond_seed"][0]),
                                     pose_scale=float(g["cond_pose_scale"][0]))
    for legacy in (False, True):
        got = _geo_check(g["points"], cond, legacy)
        assert rel_err(got.cpu(), g[f"geo_legacy{int(legacy)}"]) < 1e-5


@pytest.mark.parametrize("V,N", [(128, 64), (6890, 5000), (37, 1), (6890, 1025)])
def test_geo_features_vs_oracle(V, N):
    cond = synthetic.make_conditions(2, n_vertices=V, seed=V)
    pts = (torch.rand(2, N, 3, generator=torch.Generator().manual_seed(N)) - 0.5) * torch.tensor([1.8, 2.4, 1.0])
    _geo_check(pts, cond, False)


def test_geo_features_fp64_and_non_contiguous_conditions():
    """Conditions that need a conversion (fp64, expanded / strided views): the converted copies must outlive the launch
    (round-1 advisor finding: temporaries freed before the kernel ran could alias each other)."""
    cond = synthetic.make_conditions(2, n_vertices=777, seed=4)
    pts = (torch.rand(2, 300, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * torch.tensor([1.8, 2.4, 1.0])
    ref = O.geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"], cond["fk_matrices"],
                         cond["lbs_weights"])
    c = dev_dict(cond)
    # same-sized temporaries: vertices and tpose_vertices both fp64, skeletons a strided view, points transposed storage
    wide = torch.zeros(2, 24, 6, device=DEV)
    wide[..., ::2] = c["skeletons_xyz"]
    got = smpl.get_geo_features(dev(pts).transpose(1, 2).contiguous().transpose(1, 2), wide[..., ::2],
                                c["vertices"].double(), c["tpose_vertices"].double(), c["fk_matrices"].double(),
                                c["lbs_weights"].double())
    assert rel_err(got.cpu(), ref) < 1e-5


def test_geo_features_exact_ties_pick_first_vertex():
    cond = synthetic.make_conditions(1, n_vertices=64, seed=0, pose_scale=0.0)
    cond["vertices"][0, 10] = cond["vertices"][0, 3]          # duplicate vertex: 3 must win
    pts = cond["vertices"][:, 3:4].clone() + 1e-3
    c = dev_dict(cond)
    _, idx = smpl.get_geo_features(dev(pts), c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"],
                                   c["lbs_weights"], return_index=True)
    assert int(idx[0, 0]) == 3


def test_geo_features_ties_across_chunks_and_near_tie_overflow():
    """Filter + refine search: duplicates of the winner spread over many 64-vertex chunks (more than the eight
    remembered candidates -> whole-mesh refine), points on a sphere of equidistant vertices, far-away points."""
    V = 1500
    cond = synthetic.make_conditions(2, n_vertices=V, seed=3)
    g = torch.Generator().manual_seed(7)
    # batch 0: the same vertex repeated once per chunk, in descending chunk order of appearance
    for c in range(0, V, 64):
        cond["vertices"][0, c + 17] = cond["vertices"][0, 900]
    # batch 1: 700 vertices on a unit sphere around the origin (all within rounding of distance 1 from (0,0,0))
    d = torch.randn(700, 3, generator=g)
    cond["vertices"][1, 100:800] = d / d.norm(dim=1, keepdim=True)
    pts = (torch.rand(2, 777, 3, generator=g) - 0.5) * 2
    pts[0, :50] = cond["vertices"][0, 900] + 1e-4 * torch.randn(50, 3, generator=g)
    pts[1, :50] = 1e-3 * torch.randn(50, 3, generator=g)
    pts[:, 50:60] *= 40.0
    _geo_check(pts, cond, False)


# ------------------------------------------------------------------ A7

@pytest.mark.parametrize("shape", [(2, 5, 64, 32, 256, 128), (1, 3, 96, 48, 512, 256), (2, 4, 6, 5, 20, 12),
                                   (1, 2, 8, 4, 17, 9), (1, 1, 5, 7, 5, 7), (1, 2, 16, 16, 8, 8),
                                   # the row-caching kernel (W >= 128): ragged width / height, down-scaling, one source row
                                   (1, 2, 40, 50, 70, 130), (2, 2, 300, 200, 41, 132), (1, 3, 1, 9, 33, 257), (1, 2, 96, 96, 512, 512)])
def test_bilinear(shape):
    B, C, h, w, H, W = shape
    x = torch.randn(B, C, h, w, generator=torch.Generator().manual_seed(h * w))
    ref = torch.nn.functional.interpolate(x, (H, W), mode="bilinear")
    got = resample.bilinear_resize(dev(x), (H, W))
    assert rel_err(got.cpu(), ref) < 2e-6


@pytest.mark.parametrize("shape", [(2, 8, 6, 5, 20, 12), (1, 768, 12, 6, 64, 32), (2, 4, 5, 7, 5, 7), (1, 12, 16, 16, 8, 9), (1, 4, 1, 3, 7, 2),
                                   (3, 260, 9, 4, 50, 23)])
def test_bilinear_channels_last_forward_adjoint_and_double_backward(shape):
    """h3d_bilinear_resize_cl and its adjoint against F.interpolate (float64) and its autograd: values, gradient (up- and
    down-scaling, a single source row), and the gradient of the gradient (the adjoint's backward is the resize again)."""
    B, C, h, w, H, W = shape
    g = torch.Generator().manual_seed(h * W + C)
    x = torch.randn(B, h * w, C, generator=g)
    cot = torch.randn(B, H * W, C, generator=g)
    xr = x.double().requires_grad_()
    ref = torch.nn.functional.interpolate(xr.view(B, h, w, C).permute(0, 3, 1, 2), (H, W), mode="bilinear")
    ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, C)
    (gref,) = torch.autograd.grad(ref, xr, cot.double())
    xd = dev(x).requires_grad_()
    got = resample.bilinear_resize_cl(xd, (h, w), (H, W))
    assert rel_err(got.detach().cpu(), ref.detach()) < 2e-6
    cd = dev(cot).requires_grad_()
    (ggot,) = torch.autograd.grad(got, xd, cd, create_graph=True)
    assert rel_err(ggot.detach().cpu(), gref) < 5e-6
    # <adjoint(cot), u> differentiated w.r.t. cot is the resize of u
    u = torch.randn(B, h * w, C, generator=g)
    (back,) = torch.autograd.grad((ggot * dev(u)).sum(), cd)
    uref = torch.nn.functional.interpolate(u.double().view(B, h, w, C).permute(0, 3, 1, 2), (H, W), mode="bilinear")
    assert rel_err(back.cpu(), uref.permute(0, 2, 3, 1).reshape(B, H * W, C)) < 2e-6
    # the NCHW kernel on the same data (same association; the compiler may contract the multiply-adds differently)
    nchw = resample.bilinear_resize(dev(x).view(B, h, w, C).permute(0, 3, 1, 2).contiguous(), (H, W))
    assert rel_err(nchw.permute(0, 2, 3, 1).reshape(B, H * W, C).cpu(), got.detach().cpu()) < 1e-6


@pytest.mark.parametrize("shape", [(2, 8, 6, 5, 20, 12), (1, 768, 12, 6, 64, 32), (1, 4, 1, 3, 7, 2), (3, 260, 9, 4, 50, 23)])
def test_bilinear_channels_last_with_the_relu_inside(shape):
    """h3d_bilinear_resize_cl_relu / _relu_bwd: relu(resize(x)) in one pass and its gradient through the mask of the saved output,
    against relu(F.interpolate(..)) in float64 -- values bit-identical to relu of the plain kernel's output."""
    B, C, h, w, H, W = shape
    g = torch.Generator().manual_seed(h * W + C + 1)
    x = torch.randn(B, h * w, C, generator=g)
    cot = torch.randn(B, H * W, C, generator=g)
    xr = x.double().requires_grad_()
    ref = torch.nn.functional.interpolate(xr.view(B, h, w, C).permute(0, 3, 1, 2), (H, W), mode="bilinear")
    ref = torch.relu(ref.permute(0, 2, 3, 1).reshape(B, H * W, C))
    (gref,) = torch.autograd.grad(ref, xr, cot.double())
    xd = dev(x).requires_grad_()
    got = resample.bilinear_resize_relu_cl(xd, (h, w), (H, W))
    assert torch.equal(got.detach(), torch.relu(resample.bilinear_resize_cl(dev(x), (h, w), (H, W))))
    assert rel_err(got.detach().cpu(), ref.detach()) < 2e-6
    (ggot,) = torch.autograd.grad(got, xd, dev(cot))
    # a pixel within rounding of zero may take the other side of the mask than float64 does: compare where the reference is clear of it
    plain = resample.bilinear_resize_cl(dev(x), (h, w), (H, W)).cpu().double()
    if float((plain.abs() < 1e-6).sum()) == 0:
        assert rel_err(ggot.cpu(), gref) < 5e-6
    else:
        assert rel_err(ggot.cpu(), gref) < 1e-3


# ------------------------------------------------------------------ P1 / P2

def test_bias_act_golden():
    g = load_golden("plugin_ops")
    x, b = dev(g["bias_act_in"]["x"]), dev(g["bias_act_in"]["b"])
    for act, cases in g["bias_act"].items():
        if not isinstance(cases, dict):
            continue
        assert rel_err(bias_act_mod.bias_act(x, b, 1, act).cpu(), cases["default"]) < 2e-6, act
        got = bias_act_mod.bias_act(x, b, 1, act, alpha=0.3, gain=1.7, clamp=0.9)
        assert rel_err(got.cpu(), cases["custom"]) < 2e-6, act
    x2, b2 = dev(g["bias_act_in"]["x2"]), dev(g["bias_act_in"]["b2"])
    assert rel_err(bias_act_mod.bias_act(x2, b2, 1, "lrelu").cpu(), g["bias_act"]["dim_last"]) < 2e-6
    assert rel_err(bias_act_mod.bias_act(x2, None, act="swish").cpu(), g["bias_act"]["no_bias"]) < 2e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 2e-3), (torch.float64, 1e-7)])
def test_bias_act_dtypes_and_layouts(dtype, tol):
    g = torch.Generator().manual_seed(1)
    for shape, dim in [((4, 8, 16, 16), 1), ((3, 7), 1), ((5, 6, 3), 0), ((2, 3, 5, 4), 3), ((1024, 384), 1)]:
        x = torch.randn(shape, generator=g).to(dtype)
        b = torch.randn(shape[dim], generator=g).to(dtype)
        for act in ("linear", "lrelu", "sigmoid", "softplus", "swish", "selu", "elu", "tanh", "relu"):
            ref = O.bias_act(x.double(), b.double(), dim, act, clamp=2.0)
            got = bias_act_mod.bias_act(dev(x), dev(b), dim, act, clamp=2.0)
            assert got.dtype == dtype and got.shape == x.shape
            assert rel_err(got.cpu(), ref) < tol, (shape, act)


def test_bias_act_empty_and_errors():
    assert bias_act_mod.bias_act(torch.zeros(0, 4, device=DEV), None).numel() == 0
    with pytest.raises(KeyError):
        bias_act_mod.bias_act(torch.zeros(2, 4, device=DEV), None, act="sin")


def test_upfirdn2d_golden():
    g = load_golden("plugin_ops")
    ui = g["upfirdn_in"]
    for k, sp in g["upfirdn_specs"].items():
        f = None if sp["f"] is None else dev(ui[sp["f"]])
        out = upfirdn_mod.upfirdn2d(dev(ui["x"]), f, sp["up"], sp["down"], sp["padding"], sp["flip_filter"], sp["gain"])
        assert out.shape == g["upfirdn"][k].shape, k
        assert rel_err(out.cpu(), g["upfirdn"][k]) < 3e-6, k


def test_upfirdn2d_helpers_and_layouts():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 6, 32, 24, generator=g)
    f = upfirdn_mod.setup_filter([1, 3, 3, 1])
    assert rel_err(f, O.setup_filter([1, 3, 3, 1])) < 1e-7
    up = upfirdn_mod.upsample2d(dev(x), dev(f), up=2)
    assert up.shape == (2, 6, 64, 48)
    ref = O.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4)
    assert rel_err(up.cpu(), ref) < 3e-6
    down = upfirdn_mod.downsample2d(dev(x), dev(f), down=2)
    assert rel_err(down.cpu(), O.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1])) < 3e-6
    same = upfirdn_mod.filter2d(dev(x), dev(f))
    assert same.shape == x.shape
    assert rel_err(same.cpu(), O.upfirdn2d(x, f, padding=[2, 1, 2, 1])) < 3e-6
    # constant image stays constant in the interior (DC gain 1), round trip up->down keeps the mean
    ones = torch.ones(1, 1, 16, 16, device=DEV)
    u = upfirdn_mod.upsample2d(ones, dev(f), up=2)
    assert float((u[:, :, 4:-4, 4:-4] - 1).abs().max()) < 1e-6
    # channels_last input, fp16 and fp64
    xcl = dev(x).contiguous(memory_format=torch.channels_last)
    assert rel_err(upfirdn_mod.upsample2d(xcl, dev(f), up=2).cpu(), ref) < 3e-6
    assert rel_err(upfirdn_mod.upsample2d(dev(x).half(), dev(f), up=2).float().cpu(), ref) < 2e-3
    assert rel_err(upfirdn_mod.upsample2d(dev(x).double(), dev(f), up=2).cpu(), ref.double()) < 1e-6
    f12 = upfirdn_mod.setup_filter([1, 2, 4, 7, 9, 12, 12, 9, 7, 4, 2, 1])
    assert f12.ndim == 1
    assert rel_err(upfirdn_mod.upsample2d(dev(x), dev(f12), up=2).cpu(),
                   O.upfirdn2d(x, f12, up=2, padding=[6, 5, 6, 5], gain=4)) < 3e-6


@pytest.mark.parametrize("up,down,pad,taps", [(2, 1, (2, 1, 2, 1), 4), (1, 2, (1, 1, 1, 1), 4), (3, 2, (4, 3, 2, 5), 6), (1, 1, (-3, 2, 1, -2), 5),
                                                (2, 3, (0, 0, 7, 1), 12), (4, 1, (0, 0, 0, 0), 1)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-12), (torch.float16, 2e-3)])
def test_upfirdn2d_tiled_kernel_vs_oracle(up, down, pad, taps, dtype, tol):
    """The LDS-tiled kernel (dense NCHW) on sizes that are not multiples of its 64 x 16 tile, every resampling
    combination, negative padding, and against the per-element kernel (channels_last input takes that one)."""
    g = torch.Generator().manual_seed(up * 10 + down)
    x = torch.randn(2, 3, 37, 70, generator=g, dtype=torch.float64)
    f = torch.randn(taps, taps, generator=g)
    ref = O.upfirdn2d(x, f.double(), up=up, down=down, padding=pad, gain=1.5)
    got = upfirdn_mod.upfirdn2d(dev(x.to(dtype)), dev(f), up=up, down=down, padding=pad, gain=1.5)
    assert got.shape == ref.shape
    assert rel_err(got.cpu(), ref) < tol
    cl = upfirdn_mod.upfirdn2d(dev(x.to(dtype)).contiguous(memory_format=torch.channels_last), dev(f), up=up, down=down, padding=pad,
                               gain=1.5)
    assert rel_err(cl.cpu(), ref) < tol


@pytest.mark.parametrize("hw", [(37, 70), (33, 35), (64, 128)])
@pytest.mark.parametrize("up,down,pad", [(2, 1, (2, 1, 2, 1)), (2, 1, (1, 2, 1, 2)), (2, 1, (1, 2, 2, 1)), (2, 1, (3, 0, -1, 4)), (1, 2, (1, 1, 1, 1)),
                                         (1, 2, (2, 1, 0, 3)), (1, 1, (2, 1, 2, 1)), (1, 1, (-2, 5, 3, 0))])
def test_upfirdn2d_polyphase_kernel_vs_oracle(up, down, pad, hw):
    """The compile-time polyphase kernel (4-tap filters, 2x up / 2x down / neither, dense NCHW): every phase of the padding, sizes
    that are / are not multiples of its 64 x 64 tile and of the 4-wide vector store, 2-D and separable filters, flipped, fp16."""
    g = torch.Generator().manual_seed(up * 10 + down + hw[0])
    x = torch.randn(2, 3, *hw, generator=g, dtype=torch.float64)
    f = torch.randn(4, 4, generator=g)
    for flip in (False, True):
        ref = O.upfirdn2d(x, f.double(), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5)
        got = upfirdn_mod.upfirdn2d(dev(x.float()), dev(f), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5)
        assert got.shape == ref.shape and rel_err(got.cpu(), ref) < 1e-5
    got16 = upfirdn_mod.upfirdn2d(dev(x.half()), dev(f), up=up, down=down, padding=pad, gain=1.5)
    assert rel_err(got16.float().cpu(), O.upfirdn2d(x, f.double(), up=up, down=down, padding=pad, gain=1.5)) < 2e-3
    f1 = torch.randn(4, generator=g)                                         # separable: a row pass and a column pass
    ref = O.upfirdn2d(x, torch.outer(f1, f1).double(), up=up, down=down, padding=pad, gain=1.5)
    got = upfirdn_mod.upfirdn2d(dev(x.float()), dev(f1), up=up, down=down, padding=pad, gain=1.5)
    assert got.shape == ref.shape and rel_err(got.cpu(), ref) < 1e-5
    # a view with a storage offset that breaks the 16-byte alignment of the output is not produced by the wrapper, but an
    # input one is legal: the staging loads are scalar
    xo = dev(torch.cat([torch.zeros(1), x.float().flatten()]))[1:].view(x.shape)
    assert rel_err(upfirdn_mod.upfirdn2d(xo, dev(f), up=up, down=down, padding=pad, gain=1.5).cpu(),
                   O.upfirdn2d(x, f.double(), up=up, down=down, padding=pad, gain=1.5)) < 1e-5


@pytest.mark.parametrize("up,down,pad,taps,flip", [(2, 1, [2, 1, 2, 1], [1, 3, 3, 1], False), (1, 2, [1, 1, 1, 1], [1, 3, 3, 1], True),
                                                   (1, 1, [2, 1, 1, 2], [1, 2, 5, 1], False), (2, 2, [3, 0, 1, 2], [1, 4, 2, 1, 3], True),
                                                   ((2, 1), (1, 2), [1, 2, 2, 1], [1, 3, 3, 1], False)])
def test_upfirdn2d_gradients_vs_oracle_autograd(up, down, pad, taps, flip):
    """First and second order: dx of the op (the reference's Upfirdn2dCuda.backward = upfirdn2d with up / down swapped, the
    filter flipped, the complementary padding) and the gradient of a function of dx, against torch autograd through the
    oracle's pure-torch restatement in float64."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 11, 9, generator=g, dtype=torch.float64)
    f = torch.tensor(taps, dtype=torch.float32)
    f2 = torch.outer(f, f[: max(2, len(taps) - 1)]) / 7.0                   # asymmetric 2-D filter
    xr = x.clone().requires_grad_(True)
    ref = O.upfirdn2d(xr, f2.double(), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5)
    proj = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (gr,) = torch.autograd.grad((ref * proj).sum(), xr, create_graph=True)
    xd = dev(x.float()).requires_grad_(True)
    got = upfirdn_mod.upfirdn2d(xd, dev(f2), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5)
    assert rel_err(got.detach().cpu(), ref.detach()) < 3e-6
    (gg,) = torch.autograd.grad((got * dev(proj.float())).sum(), xd, create_graph=True)
    assert gg.shape == xd.shape and rel_err(gg.detach().cpu(), gr.detach()) < 1e-5
    # double backward proper: d/d(proj) of ||dx||^2 -- runs the op's backward-of-backward
    pd = dev(proj.float()).requires_grad_(True)
    (g1,) = torch.autograd.grad((upfirdn_mod.upfirdn2d(xd, dev(f2), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5) * pd).sum(),
                                xd, create_graph=True)
    (gp,) = torch.autograd.grad((g1 * g1).sum(), pd)
    pr = proj.clone().requires_grad_(True)
    (r1,) = torch.autograd.grad((O.upfirdn2d(xr, f2.double(), up=up, down=down, padding=pad, flip_filter=flip, gain=1.5) * pr).sum(),
                                xr, create_graph=True)
    (rp,) = torch.autograd.grad((r1 * r1).sum(), pr)
    assert rel_err(gp.cpu(), rp) < 1e-5
