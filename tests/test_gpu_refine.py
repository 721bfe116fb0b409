"""Refinement of ill-conditioned last samples (round 6; h3d_render_fused_x2_geo_ref + h3d_render_fused_x3_geo_units).

The reference gives a ray's last sample delta = 1e9 (lib/generators/volume_rendering.py:21): its alpha is 0 or 1 by the SIGN of its
density, and the background term (last_back / white_back, :38-49) flips by the ray's whole remaining transmittance.  The x2 render
lists the wave units holding a ray whose last density lies within eps of zero and the three-product engine redoes exactly those:
the listed rays are bit-identical to the x3 engine's, every other ray keeps its x2 values."""
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


def setup(S, R, hidden, B=2, seed=3, sigma_gain=40.0):
    state, net = random_state(hidden, hidden, seed=seed, precision="f16x2")
    with torch.no_grad():
        net.sigma_layer.weight.mul_(sigma_gain)
        state["neural_field.sigma_layer.weight"] = net.sigma_layer.weight.detach().cpu().clone()
    N, V = R * S, 300
    cond = synthetic.make_conditions(B, n_vertices=V, seed=seed)
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(B, N, 3, generator=g) * 2 - 1
    freq = torch.randn(B, 4 * hidden, generator=g) * 0.5
    phase = torch.randn(B, 4 * hidden, generator=g)
    z = torch.sort(torch.rand(B, R, S, 1, generator=g) + 11, dim=2).values
    c = {k: v.to(DEV) for k, v in cond.items()}
    vik = smpl.vertex_inverse_transforms(c["fk_matrices"], c["lbs_weights"])
    idx = smpl.nearest_vertex(pts.to(DEV), c["vertices"])

    def run():
        return net.render_geo(pts.to(DEV), freq.to(DEV), phase.to(DEV), idx, c["skeletons_xyz"], c["vertices"], c["tpose_vertices"],
                              vik, None, z.to(DEV), S, input_scaler=0.7, clamp_mode="relu", last_back=True, white_back=True)

    def oracle():
        geo = O.geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"], cond["fk_matrices"],
                             cond["lbs_weights"], False)
        dirs = torch.zeros(B, N, 3)
        dirs[..., 2] = -1
        sd = {k: v.double() for k, v in state.items()}
        field = O.neural_field(sd, pts.double(), freq.double(), phase.double(), geo.double(), dirs.double(), 0.7)
        return field.reshape(B, R, S, -1), O.ray_integration(field.reshape(B, R, S, -1), z.double(), None, "relu", True, True)

    return net, run, oracle


@pytest.mark.parametrize("S,R,hidden", [(64, 40, 256), (128, 12, 128), (16, 64, 64), (32, 33, 64)])
def test_listed_units_are_the_x3_engines_and_the_others_keep_their_x2_values(S, R, hidden):
    net, run, _ = setup(S, R, hidden)
    B = 2
    unit = max(S, 32)
    n_units = (R * S + unit - 1) // unit
    net.refine_last_sample = False
    x2 = [t.clone() for t in run()]
    net.precision = "f16x3"
    x3 = [t.clone() for t in run()]
    net.precision = "f16x2"
    assert not torch.equal(x2[0], x3[0])
    # eps = 0: nothing listed (a density of exactly zero aside), the x2 image untouched
    net.refine_last_sample, net.refine_eps = True, 0.0
    out = run()
    assert int(net.refined_units().sum()) == 0 and all(torch.equal(a, b) for a, b in zip(out, x2))
    # eps huge: EVERY unit is listed; the first `cap` arrivals are redone -- with cap >= the unit count that is the whole x3 render
    net.refine_eps, net.refine_capacity = 1e30, max(8, n_units)
    out = run()
    assert net.refined_units().tolist() == [n_units] * B
    assert all(torch.equal(a, b) for a, b in zip(out, x3))
    # capacity below the number of listed units: exactly `cap` units per item are the x3 engine's, the rest the x2 engine's
    cap = max(1, n_units // 3)
    net.refine_capacity = cap
    out = run()
    assert net.refined_units().tolist() == [n_units] * B              # the count keeps counting beyond the capacity
    rays_per_unit = unit // S
    f = out[0].view(B, -1, out[0].shape[-1])
    is3 = (f == x3[0].view_as(f)).all(-1)
    is2 = (f == x2[0].view_as(f)).all(-1)
    assert bool((is3 | is2).all())
    for b in range(B):
        units3 = is3[b, : n_units * rays_per_unit].view(n_units, rays_per_unit).all(-1) & ~is2[b, : n_units * rays_per_unit].view(n_units, rays_per_unit).all(-1)
        assert int(units3.sum()) == cap, (int(units3.sum()), cap)


def test_rays_with_a_last_density_near_zero_take_the_x3_sign():
    """The default threshold on a field whose densities straddle zero: the rays the ORACLE calls ill-conditioned for the x2
    arithmetic (|sigma_last| <= 1e-3 of the largest density) are the x3 engine's bit for bit -- whatever their sign flip does to
    the background term, x2-with-refinement and x3 agree on it -- and every listed ray is one."""
    S, R, hidden = 64, 4000, 256
    net, run, oracle = setup(S, R, hidden, B=1, seed=9, sigma_gain=40.0)
    assert net.refine_last_sample and net.refine_eps == 1e-3
    out = run()
    n_listed = int(net.refined_units()[0])
    net.refine_last_sample = False
    x2 = run()
    net.precision = "f16x3"
    x3 = run()
    field, ref = oracle()
    sigma = field[..., -1]                                               # [1, R, S]
    scale = float(sigma.abs().max())
    near = (sigma[0, :, -1].abs() <= 2e-4 * scale)                       # well inside the listing band (eps 1e-3 of >= the ray's scale... see below)
    f, f2, f3 = out[0][0].cpu(), x2[0][0].cpu(), x3[0][0].cpu()
    same3 = (f == f3).all(-1)
    print(f"{n_listed} units listed of {R}; {int(near.sum())} rays within 2e-4 of zero; x2 vs x3 flips: "
          f"{int(((f2 - f3).abs().amax(-1) > 1e-2).sum())}, refined vs x3 flips: {int(((f - f3).abs().amax(-1) > 1e-2).sum())}")
    assert 0 < n_listed < R // 20
    # the band is relative to max(ray's largest |density|, ||w_sigma||) and the field's largest |density| is ~3 ||w_sigma|| (inputs
    # are sines): |sigma_last| <= 2e-4 * max|sigma| of the whole field lies safely inside it
    assert scale <= 5.0 * float(net._sigma_scale(DEV))
    listed_ok = same3[near]
    assert bool(listed_ok.all()), "a ray with a last density near zero was not refined"
    assert int(same3.sum()) == n_listed                                  # S > 32: a unit is a ray; exactly the listed ones changed
    assert int(((f - f3).abs().amax(-1) > 1e-2).sum()) == 0              # no O(1) disagreement with the three-product engine left
    assert rel_err(f[~same3], f2[~same3]) == 0.0
