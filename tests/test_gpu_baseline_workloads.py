"""The exact BASELINE.json workloads on the GPU against the CPU oracle restricted to a subset of pixels / rays
(oracle/h3d_oracle.py: generator_forward_subset, pinned to the reference's vectors by tests/test_oracle_golden.py), plus
batch consistency (image i of the batch == the batch-1 run of sample i).

cfg 2: MAP3DBN  (Hd 384), 256x256 from 64x64 rays x 32, batch 8
cfg 3: MAP3DBN512 (Hd 256), 512x512 from 96x96 rays x 64, batch 16       <- the bench.py workload
cfg 3L: MAP3DBN512L (Hd 420, legacy / isolated: the released checkpoint's architecture), same geometry, batch 4
cfg 5: MAP3DBN512, 1024x1024 from 192x192 rays x 128, batch 4
"""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import rel_err, rel_err_channels, rel_err_rms

pytestmark = pytest.mark.gpu
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")
configs = importlib.import_module("3dhumangan_amd.configs")
_lib = importlib.import_module("3dhumangan_amd._lib")
DEV = "cuda"
TOL = 1e-3          # north_star: within 1e-3 relative of the reference CPU path


def make(cfg_name, gen_hw, render_hw, S, B, seed):
    cfg = {k: v for k, v in getattr(configs, cfg_name).items() if isinstance(k, str)}
    cfg.update(gen_height=gen_hw[0], gen_width=gen_hw[1], render_height=render_hw[0], render_width=render_hw[1],
               num_steps=S, dataset_length=4, nerf_noise=0, last_back=cfg["eval_last_back"])
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(seed)
    G = gens.Map3DGenerator(**cfg).to(DEV).eval()
    G.set_device(DEV)
    g = torch.Generator().manual_seed(seed)
    cond = synthetic.make_conditions(B, 6890, seed=seed % 1000)
    z = torch.randn(B, cfg["latent_dim"], generator=g)
    jit = torch.rand(B, render_hw[0] * render_hw[1], S, 1, generator=g)
    return G, cfg, z, cond, jit


def cells_for(render_hw, n, seed):
    Hr, Wr = render_hw
    g = torch.Generator().manual_seed(seed)
    cells = [(0, 0), (0, Wr - 1), (Hr - 1, 0), (Hr - 1, Wr - 1), (Hr // 2, Wr // 2)]           # clamped edges + centre
    cy = torch.randint(0, Hr, (n,), generator=g).tolist()
    cx = torch.randint(0, Wr, (n,), generator=g).tolist()
    return cells + list(zip(cy, cx))


def check_workload(cfg_name, gen_hw, render_hw, S, B, oracle_items, n_cells=24, seed=21):
    G, cfg, z, cond, jit = make(cfg_name, gen_hw, render_hw, S, B, seed)
    cd = {k: v.to(DEV) for k, v in cond.items()}
    out = G.forward(z.to(DEV), cd, jitter=jit.to(DEV), **cfg)
    rgb, ren = out["rgbs"].cpu(), out["rgbs_render"].cpu()
    plan = G.synthesis_plan(DEV)
    redone = plan.x2_fallback_items()             # range guard / sampled error monitor of the x2 engine: the items redone on x3
    if plan.x2_monitor_errors() is not None:
        print(f"{cfg_name} {gen_hw} B={B}: x2 monitor sampled errors {[round(float(v), 6) for v in plan.x2_monitor_errors().cpu()]}, "
              f"items redone on x3: {redone}")
    assert rgb.shape == (B, 3) + tuple(gen_hw) and torch.isfinite(rgb).all()
    # ---- oracle on a subset of pixels (and the rays they need) for a few batch items
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    ocfg = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
    pix = O.pixels_of_cells(cells_for(render_hw, n_cells, seed), gen_hw, render_hw)
    assert len(pix) > 100
    for i in oracle_items:
        ci = {k: v[i:i + 1] for k, v in cond.items()}
        ref = O.generator_forward_subset(sd, ocfg, z[i:i + 1], ci, jit[i:i + 1], pix)
        got = rgb[i:i + 1].flatten(2)[:, :, pix]
        got_r = ren[i:i + 1].flatten(2)[:, :, ref["ray_subset"]]
        e = (rel_err(got, ref["rgbs"]), rel_err_channels(got, ref["rgbs"]), rel_err_rms(got, ref["rgbs"]),
             rel_err_channels(got_r, ref["rgbs_render"]))
        print(f"{cfg_name} {gen_hw} B={B} item {i}: image max {e[0]:.2e} per-channel {e[1]:.2e} rms {e[2]:.2e}; "
              f"render per-channel {e[3]:.2e}  ({len(pix)} pixels, {len(ref['ray_subset'])} rays)")
        assert max(e) < TOL, e
    # ---- batch consistency: sample i alone gives the same image
    for i in sorted(set([0, B - 1] + list(oracle_items))):
        ci = {k: v[i:i + 1].to(DEV) for k, v in cond.items()}
        one = G.forward(z[i:i + 1].to(DEV), ci, jitter=jit[i:i + 1].to(DEV), **cfg)
        # round 6: the guard / monitor flags are per item, so an item takes the same engine alone and in its batch and its image
        # does not depend on its batch mates (an item whose sampled error sits within rounding of the tolerance could still differ
        # in that decision: the two runs' table GEMMs are not bit-identical)
        same_engine = plan.x2_fell_back() == (i in redone)
        assert same_engine or abs(float(plan.x2_monitor_errors()[0]) - plan.x2_monitor_tol) < 1e-5
        assert rel_err_channels(one["rgbs"].cpu(), rgb[i:i + 1]) < (2e-5 if same_engine else 1e-3)
        assert rel_err_channels(one["rgbs_render"].cpu(), ren[i:i + 1]) < 2e-5
    return G


def test_cfg3_bench_workload_b16_512sq():
    """BASELINE config 3 exactly as bench.py times it: MAP3DBN512, 512x512, 96x96 rays, 64 samples, batch 16."""
    G = check_workload("MAP3DBN512", (512, 512), (96, 96), 64, 16, oracle_items=(0, 9, 15))
    assert G.neural_field.precision == "f16x2" and G.synthesis_plan(DEV).engine == "f16x2"       # what bench.py times


def test_cfg3_native_aspect_b16():
    check_workload("MAP3DBN512", (512, 256), (96, 48), 64, 16, oracle_items=(3,))


def test_cfg5_1024sq_s128_b4():
    """BASELINE config 5 geometry: 1024x1024, 192x192 rays, 128 samples per ray, batch 4."""
    assert _lib.load().h3d_synthesis_x3_geometry_ok(1024, 1024, 192, 192) == 1
    check_workload("MAP3DBN512", (1024, 1024), (192, 192), 128, 4, oracle_items=(0, 3), n_cells=16)


def test_cfg2_256sq_b8():
    """BASELINE config 2: MAP3DBN (hidden 384), 256x256 square from 64x64 rays x 32, batch 8."""
    check_workload("MAP3DBN", (256, 256), (64, 64), 32, 8, oracle_items=(0, 7))


def test_cfg3L_released_checkpoint_architecture_b4():
    """MAP3DBN512L (hidden 420, legacy geometry-feature order, isolated styles) at the cfg-3 geometry."""
    check_workload("MAP3DBN512L", (512, 512), (96, 96), 64, 4, oracle_items=(1,))


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_cfg3_bench_workload_more_seeds_checked_as_the_bench_checks_itself(seed):
    """Three more (weights, latents, pose, jitter) draws of the bench workload through bench.py's own in-run check (self_check: the
    oracle on a pixel / ray subset, errors relative to the whole image's channel maximum, rays excluded only where the ORACLE's
    last-sample density is within the tolerance of zero, at most max(1, 5e-4 n) of them): the default engines with their guard and
    monitor stay inside 1e-3 on draws the round's measurements were not tuned on."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    G, cfg, z, cond, jit = make("MAP3DBN512", (512, 512), (96, 96), 64, 16, seed)
    chk = bench.self_check(G, cfg, z.to(DEV), {k: v.to(DEV) for k, v in cond.items()}, jit.to(DEV), [0, 11])
    assert chk["pixel_fraction"] >= 0.05
    print(f"seed {seed}: image {chk['max_rel_err']:.2e} (image-normalised {chk['max_rel_err_image_norm']:.2e}), render "
          f"{chk['max_rel_err_render']:.2e}, excluded rays {chk['rays_excluded_as_ill_conditioned_in_the_oracle']}, monitor {chk['x2_monitor']}")
    assert chk["ok"] and chk["max_rel_err"] < TOL and chk["max_rel_err_render"] < TOL
    assert chk["synthesis_engine"] == "f16x2"
