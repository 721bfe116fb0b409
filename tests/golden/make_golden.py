"""Generate the golden fixtures in this directory from the real Python reference.

BUILD CONTAINER ONLY: imports /root/reference through the shims of
``_ref_import.py`` (missing third-party packages; functional kNN stand-in),
drives the reference modules with seeded synthetic inputs and stores *data*
(inputs, weights, every random tensor the reference drew, outputs) as .npz.
Nothing from the reference's source travels.  Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import _ref_import  # noqa: E402

_ref_import.install()

import configs as ref_configs  # noqa: E402
import lib.generators.map3d_generator as ref_gen  # noqa: E402
import lib.generators.volume_rendering as ref_vr  # noqa: E402
from lib import implicit_funcitions as ref_impl  # noqa: E402
from lib.components import smpl as ref_smpl  # noqa: E402
from lib.components import map3d_layers as ref_layers  # noqa: E402
from lib.components import cips_layers as ref_cips  # noqa: E402
from lib.components.ops import bias_act as ref_bias_act  # noqa: E402
from lib.components.ops import upfirdn2d as ref_upfirdn2d  # noqa: E402

synthetic = importlib.import_module("3dhumangan_amd.synthetic")


def np_(t):
    return t.detach().cpu().numpy()


def tiny_cfg(**over):
    cfg = {k: v for k, v in ref_configs.MAP3DBN.items() if isinstance(k, str)}
    cfg.update(hidden_dim=32, latent_dim=32, feature_dim=32, gen_height=16, gen_width=8,
               render_height=8, render_width=4, num_steps=8, dataset_length=4)
    cfg.update(over)
    cfg["neural_field_cls"] = ref_impl.COORDCONCATSIREN
    return cfg


def condition_weights(G, seed):
    """Make eval-mode numerics sane and non-trivial: exact spectral-norm u/v,
    randomised BN statistics/affine, non-zero biases."""
    g = torch.Generator().manual_seed(seed)
    sd = G.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith("weight_orig"):
                U, S, Vh = torch.linalg.svd(v.flatten(1), full_matrices=False)
                sd[k.replace("weight_orig", "weight_u")].copy_(U[:, 0])
                sd[k.replace("weight_orig", "weight_v")].copy_(Vh[0])
            elif k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.2)
            elif k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif "first_norm.weight" in k:
                v.copy_(1 + 0.2 * torch.randn(v.shape, generator=g))
            elif k.endswith("bias") and v.abs().max() == 0:
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
        # make the densities matter (a fresh SIREN has sigma ~ 0.05, i.e. nearly empty space)
        sd["neural_field.sigma_layer.weight"].mul_(60.0)
        sd["neural_field.sigma_layer.bias"].fill_(0.5)
    G.load_state_dict(sd)
    # perturb u slightly off the singular vector on one layer so sigma != s_max exactly
    return G


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np_(vv) if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = np_(v) if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(path, **flat)
    print(f"{name}.npz  {os.path.getsize(path) / 1e6:.2f} MB  ({len(flat)} arrays)")


def generator_fixture(name, seed, n_vertices=128, batch=2, nerf_noise=0.3, **over):
    cfg = tiny_cfg(**over)
    torch.manual_seed(seed)
    G = ref_gen.Map3DGenerator(**cfg).eval()
    G.set_device("cpu")
    condition_weights(G, seed)
    cond = synthetic.make_conditions(batch, n_vertices=n_vertices, seed=seed, pose_scale=0.6)
    z = torch.randn(batch, cfg["latent_dim"], generator=torch.Generator().manual_seed(seed + 1))
    run = dict(cfg)
    run["nerf_noise"] = nerf_noise
    R, S = cfg["render_height"] * cfg["render_width"], cfg["num_steps"]

    # ---- whole forward, RNG replayed in the reference's consumption order
    rs = seed + 7
    torch.manual_seed(rs)
    jitter = torch.rand(batch, R, S, 1)
    torch.randn(batch, 1), torch.randn(batch, 1)
    noise = torch.randn(batch, R, S, 1) * nerf_noise
    torch.manual_seed(rs)
    with torch.no_grad():
        out = G.forward(z, cond, **run)

    # ---- stage by stage with the same random tensors
    with torch.no_grad():
        zin = z if cfg.get("neural_field_latent_input", True) else torch.zeros_like(z)
        freq, phase = G.neural_field_mapping_network(zin)
        _, styles = G.synthesis_mapping_network(z)
        focals = cond["intrinsics"][:, 0, 0]
        pc, zv, dc = ref_vr.get_initial_rays_weak_perspective(
            focals, cond["scales"], S, resolution=(cfg["render_width"], cfg["render_height"]), device="cpu",
            ray_start=cfg["ray_start"], ray_end=cfg["ray_end"])
        torch.manual_seed(rs)
        pts, zv2, dirs, _, _, _, _ = ref_vr.transform_sampled_points(
            pc, zv, dc, cam2world_matrix=cond["cam2world_matrices"], device="cpu", mode=cfg["sample_dist"])
        pts = pts.reshape(batch, R * S, 3)
        dirs_exp = ref_vr.expand_ray_directions(dirs, S)
        if cfg["lock_view_dependence"]:
            dirs_exp = torch.zeros_like(dirs_exp)
            dirs_exp[..., -1] = -1
        geo = G.get_geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"],
                                 cond["fk_matrices"], cond["lbs_weights"])
        field = G.neural_field(pts, freq, phase, geo, ray_directions=dirs_exp,
                               input_scaler=2. / cfg["side_length"], modulation_scaler=1.)
        field = field.reshape(batch, R, S, -1)
        # ray_integration draws randn itself: replay
        torch.manual_seed(rs)
        torch.rand(batch, R, S, 1), torch.randn(batch, 1), torch.randn(batch, 1)
        feats, depth, weights = ref_vr.ray_integration(
            field, zv2, device="cpu", white_back=cfg.get("white_back", False), last_back=cfg.get("last_back", False),
            clamp_mode=cfg["clamp_mode"], noise_std=nerf_noise)

    # ---- staged_forward with truncation (draws randn(10000, L) first)
    torch.manual_seed(rs + 1)
    torch.randn(10000, cfg["latent_dim"])
    jitter_s = torch.rand(batch, R, S, 1)
    torch.manual_seed(rs + 1)
    srun = dict(run)
    srun.update(truncation_psi=0.7, nerf_noise=0, last_back=cfg["eval_last_back"], return_internal=True)
    with torch.no_grad():
        sout = G.staged_forward(z, cond, **srun)
    avg = dict(z=G.avg_latent[0], freq=G.avg_latent[1], phase=G.avg_latent[2], styles=G.avg_latent[3])

    meta = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool))}
    meta["mod_blocks"] = list(cfg["mod_blocks"])
    save(name,
         state=G.state_dict(), cond=cond, z=z, jitter=jitter, noise=noise,
         meta_json=np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8),
         out=dict(rgbs=out["rgbs"], rgbs_render=out["rgbs_render"]),
         stage=dict(freq=freq, phase=phase, styles=styles, points=pts, z_vals=zv2, dirs=dirs_exp, geo=geo,
                    field=field, feats=feats, depth=depth, weights=weights),
         staged=dict(jitter=jitter_s, rgbs=sout["rgbs"], rgbs_render=sout["rgbs_render"], depths=sout["depths"],
                     m3d_2_feature_map=sout["m3d_2_feature_map"], m3d_5_rgb=sout["m3d_5_rgb"],
                     m3d_8_feature_map=sout["m3d_8_feature_map"]),
         avg=avg)


def hierarchical_fixture(name="gen_tiny_hierarchical", seed=5, n_vertices=128, batch=2, nerf_noise=0.3):
    """forward with hierarchical_sample=True (map3d_generator.py:449-516) + sample_pdf alone (volume_rendering.py:261-303),
    every random tensor replayed in the reference's consumption order: jitter, two unused randn, the coarse
    ray_integration's noise, sample_pdf's uniforms, the final ray_integration's noise."""
    cfg = tiny_cfg(hierarchical_sample=True)
    torch.manual_seed(seed)
    G = ref_gen.Map3DGenerator(**cfg).eval()
    G.set_device("cpu")
    condition_weights(G, seed)
    cond = synthetic.make_conditions(batch, n_vertices=n_vertices, seed=seed, pose_scale=0.6)
    z = torch.randn(batch, cfg["latent_dim"], generator=torch.Generator().manual_seed(seed + 1))
    run = dict(cfg)
    run["nerf_noise"] = nerf_noise
    R, S = cfg["render_height"] * cfg["render_width"], cfg["num_steps"]
    rs = seed + 7
    torch.manual_seed(rs)
    jitter = torch.rand(batch, R, S, 1)
    torch.randn(batch, 1), torch.randn(batch, 1)
    noise_coarse = torch.randn(batch, R, S, 1) * nerf_noise
    u = torch.rand(batch * R, S)
    noise = torch.randn(batch, R, 2 * S, 1) * nerf_noise
    torch.manual_seed(rs)
    with torch.no_grad():
        out = G.forward(z, cond, **run)
    # sample_pdf alone: ragged weights incl. exact zeros (bins that can never be sampled) and a degenerate all-zero ray
    g = torch.Generator().manual_seed(seed + 11)
    bins = torch.sort(torch.rand(6, 13, generator=g) * 2 + 10, dim=1).values
    w = torch.rand(6, 12, generator=g)
    w[1, 3:7] = 0
    w[2] = 0
    w[3, :-1] = 0
    torch.manual_seed(rs + 3)
    u2 = torch.rand(6, 20)
    torch.manual_seed(rs + 3)
    samples = ref_vr.sample_pdf(bins, w, 20, det=False)
    meta = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool))}
    meta["mod_blocks"] = list(cfg["mod_blocks"])
    save(name, state=G.state_dict(), cond=cond, z=z, jitter=jitter, noise_coarse=noise_coarse, u=u, noise=noise,
         meta_json=np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8),
         out=dict(rgbs=out["rgbs"], rgbs_render=out["rgbs_render"]),
         pdf=dict(bins=bins, weights=w, u=u2, samples=samples))


def field_fixture(name, seed, hidden, n_points=96, half_exact=False):
    """COORDCONCATSIREN alone at a width that is / is not a multiple of the MFMA tile.  half_exact: the weights are rounded to
    fp16-representable values first and stored as fp16 (the real-width fixtures: 1 - 2.5 MB instead of 2 - 5)."""
    torch.manual_seed(seed)
    net = ref_impl.COORDCONCATSIREN(input_dim=3, latent_dim=hidden, hidden_dim=hidden, geo_feature_dim=31,
                                    output_dim=hidden + 4, feature_dim=hidden, num_blocks=4).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
            if half_exact:
                p.copy_(p.half().float())
    B = 2
    pts = torch.rand(B, n_points, 3) * 2 - 1
    geo = torch.rand(B, n_points, 31) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(B, n_points, 3), dim=-1)
    freq = torch.randn(B, 4 * hidden) * 0.5
    phase = torch.randn(B, 4 * hidden)
    with torch.no_grad():
        out = net(pts, freq, phase, geo, dirs, input_scaler=2. / 2.85)
    state = {"neural_field." + k: (v.half() if half_exact else v) for k, v in net.state_dict().items()}
    save(name, state=state, points=pts, geo=geo, dirs=dirs, freq=freq, phase=phase, out=out)


def integration_fixture():
    g = torch.Generator().manual_seed(5)
    cases = {}
    for i, (S, C, clamp, last_back, white_back) in enumerate(
            [(32, 11, "relu", False, True), (64, 7, "softplus", True, False), (128, 5, "relu", True, True),
             (8, 35, "relu", False, False)]):
        B, R = 2, 24
        field = torch.randn(B, R, S, C + 1, generator=g)
        field[..., -1] = field[..., -1] * 20 - 4
        z = torch.sort(torch.rand(B, R, S, 1, generator=g) + 10, dim=2).values
        torch.manual_seed(100 + i)
        noise = torch.randn(B, R, S, 1) * 0.5
        torch.manual_seed(100 + i)
        f, d, w = ref_vr.ray_integration(field.clone(), z, device="cpu", noise_std=0.5, last_back=last_back,
                                         white_back=white_back, clamp_mode=clamp)
        cases[f"c{i}"] = dict(field=field, z=z, noise=noise, feats=f, depth=d, weights=w,
                              flags=np.array([S, C, clamp == "softplus", last_back, white_back]))
    save("ray_integration", **cases)


def geo_fixture():
    cond = synthetic.make_conditions(2, n_vertices=6890, seed=3, pose_scale=0.7)
    g = torch.Generator().manual_seed(9)
    pts = (torch.rand(2, 192, 3, generator=g) - 0.5) * torch.tensor([1.6, 2.2, 0.8])
    outs = {}
    for legacy in (False, True):
        with torch.no_grad():
            outs[f"geo_legacy{int(legacy)}"] = ref_smpl.get_geo_features(
                pts.clone(), cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"], cond["fk_matrices"],
                cond["lbs_weights"], legacy)
    # conditions are regenerated from the seed in the test (synthetic.make_conditions is product code)
    save("geo_features", points=pts, cond_seed=np.array([3]), cond_pose_scale=np.array([0.7]),
         vertices_checksum=cond["vertices"].double().sum(), **outs)


def ops_fixture():
    g = torch.Generator().manual_seed(11)
    cases = {}
    x = torch.randn(3, 5, 4, 6, generator=g) * 2
    b = torch.randn(5, generator=g)
    for act in ref_bias_act.activation_funcs:
        if act == "sin":
            continue
        cases[f"bias_act/{act}/default"] = ref_bias_act.bias_act(x, b, dim=1, act=act)
        cases[f"bias_act/{act}/custom"] = ref_bias_act.bias_act(x, b, dim=1, act=act, alpha=0.3, gain=1.7, clamp=0.9)
    x2 = torch.randn(7, 6, generator=g)
    b2 = torch.randn(6, generator=g)
    cases["bias_act/dim_last"] = ref_bias_act.bias_act(x2, b2, dim=1, act="lrelu")
    cases["bias_act/no_bias"] = ref_bias_act.bias_act(x2, None, act="swish")
    cases["bias_act_in/x"], cases["bias_act_in/b"], cases["bias_act_in/x2"], cases["bias_act_in/b2"] = x, b, x2, b2

    xi = torch.randn(2, 3, 9, 7, generator=g)
    f4 = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    f12 = ref_upfirdn2d.setup_filter([1, 2, 4, 7, 9, 12, 12, 9, 7, 4, 2, 1.5])      # 1-D separable, asymmetric
    f2d = torch.rand(3, 5, generator=g)
    cases["upfirdn_in/x"], cases["upfirdn_in/f4"], cases["upfirdn_in/f12"], cases["upfirdn_in/f2d"] = xi, f4, f12, f2d
    specs = {
        "up2": dict(f="f4", up=2, down=1, padding=[2, 1, 2, 1], flip_filter=False, gain=4),
        "down2": dict(f="f4", up=1, down=2, padding=[1, 1, 1, 1], flip_filter=False, gain=1),
        "filt": dict(f="f4", up=1, down=1, padding=[1, 2, 1, 2], flip_filter=True, gain=1),
        "sep12_up2": dict(f="f12", up=2, down=1, padding=[6, 5, 6, 5], flip_filter=False, gain=4),
        "sep12_flip": dict(f="f12", up=1, down=1, padding=[5, 6, 5, 6], flip_filter=True, gain=1),
        "asym": dict(f="f2d", up=[2, 1], down=[1, 2], padding=[3, 0, 1, 2], flip_filter=False, gain=1.5),
        "negpad": dict(f="f4", up=2, down=2, padding=[-1, 2, 3, -2], flip_filter=False, gain=1),
        "nofilt": dict(f=None, up=2, down=1, padding=0, flip_filter=False, gain=1),
    }
    for k, sp in specs.items():
        fil = {"f4": f4, "f12": f12, "f2d": f2d, None: None}[sp["f"]]
        cases[f"upfirdn/{k}"] = ref_upfirdn2d.upfirdn2d(xi, fil, up=sp["up"], down=sp["down"], padding=sp["padding"],
                                                      flip_filter=sp["flip_filter"], gain=sp["gain"])
    cases["upfirdn_specs_json"] = np.frombuffer(__import__("json").dumps(specs).encode(), dtype=np.uint8)

    torch.manual_seed(21)
    m = ref_layers.SpatialStyleModLayer(in_channel=24, out_channel=40, style_dim=16)
    with torch.no_grad():
        m.bias.add_(0.1 * torch.randn_like(m.bias))
    xm, sm = torch.randn(2, 50, 24), torch.randn(2, 50, 16)
    with torch.no_grad():
        cases["modconv1x1/out"] = m(xm, sm)
    cases["modconv1x1/x"], cases["modconv1x1/style"] = xm, sm
    for k, v in m.state_dict().items():
        cases[f"modconv1x1/state/{k}"] = v
    for ks in (1, 3):
        c = ref_cips.StyleModLayer(in_channel=12, out_channel=20, kernel_size=ks, style_dim=10)
        with torch.no_grad():
            c.bias.add_(0.1 * torch.randn_like(c.bias))
        xc, sc = torch.randn(2, 12, 9, 6), torch.randn(2, 10)
        with torch.no_grad():
            cases[f"modconv2d_k{ks}/out"] = c(xc, sc)
        cases[f"modconv2d_k{ks}/x"], cases[f"modconv2d_k{ks}/style"] = xc, sc
        for k, v in c.state_dict().items():
            cases[f"modconv2d_k{ks}/state/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "plugin_ops.npz"),
                        **{k: (np_(v) if torch.is_tensor(v) else v) for k, v in cases.items()})
    print("plugin_ops.npz", len(cases), "arrays")


def config_fixture():
    import json
    out = {}
    for name in ("MAP3DBN", "MAP3DBN512", "MAP3DBN512L"):
        cfg = getattr(ref_configs, name)
        out[name] = {("step:%d" % k if isinstance(k, int) else k): (v.__name__ if isinstance(v, type) else v)
                     for k, v in cfg.items()}
    meta = {}
    for name, step in (("MAP3DBN", 0), ("MAP3DBN", 150000), ("MAP3DBN", 400000), ("MAP3DBN512L", 7)):
        m = ref_configs.extract_metadata(getattr(ref_configs, name), step)
        meta[f"{name}@{step}"] = {k: (v.__name__ if isinstance(v, type) else v) for k, v in m.items()}
    steps = {}
    for name in ("MAP3DBN", "MAP3DBN512L"):
        for step in (0, 5, 140001, 200000, 300001):
            nxt = ref_configs.next_upsample_step(getattr(ref_configs, name), step)
            steps[f"{name}@{step}"] = [None if nxt == float("Inf") else nxt,
                                       ref_configs.last_upsample_step(getattr(ref_configs, name), step)]
    with open(os.path.join(HERE, "configs.json"), "w") as f:
        json.dump({"configs": out, "extract_metadata": meta, "upsample_steps": steps}, f, indent=1, sort_keys=True,
                  default=list)
    print("configs.json")


def harness_fixture():
    """apps/sample_from_generator.py:generate_frames of the REFERENCE, driven with the shared stubs."""
    import apps.sample_from_generator as ref_app
    from _stub_generator import StubGenerator, StubPreprocessor
    cfg = dict(latent_dim=16, gen_height=12, gen_width=6)
    cond = {"dummy": torch.arange(6.0).view(1, 6)}
    out = {}
    for baf in (0, 1):
        G = StubGenerator()
        frames, sem = ref_app.generate_frames(G, StubPreprocessor(), cfg, 7, cond, 5, 0.5, 0.2, bool(baf))
        out[f"frames{baf}"] = frames
        out[f"sem{baf}"] = sem
        out[f"z{baf}"] = torch.cat([c[0] for c in G.calls])
        out[f"c2w{baf}"] = torch.cat([c[1] for c in G.calls])
    save("app_harness", **out)


if __name__ == "__main__":
    if "--only-real-width" in sys.argv:
        for hidden in (256, 384, 420):
            field_fixture(f"field_h{hidden}", seed=100 + hidden, hidden=hidden, n_points=32, half_exact=True)
        sys.exit(0)
    config_fixture()
    harness_fixture()
    hierarchical_fixture()
    generator_fixture("gen_tiny_mixed", seed=1)
    generator_fixture("gen_tiny_isolated_legacy", seed=2, map3d_mode="isolated", legacy_mode=True,
                      last_back=True, clamp_mode="softplus", hidden_dim=48, latent_dim=48, feature_dim=48,
                      render_height=6, render_width=5, gen_height=20, gen_width=12, num_steps=16, nerf_noise=0.0)
    field_fixture("field_h64", seed=3, hidden=64)
    field_fixture("field_h40", seed=4, hidden=40)
    for hidden in (256, 384, 420):               # the three shipped widths (MAP3DBN512, MAP3DBN, MAP3DBN512L), 64 points each
        field_fixture(f"field_h{hidden}", seed=100 + hidden, hidden=hidden, n_points=32, half_exact=True)
    integration_fixture()
    geo_fixture()
    ops_fixture()
