"""Ad-hoc per-kernel timing on one GPU (development aid; bench.py is the contract)."""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vr = importlib.import_module("3dhumangan_amd.lib.generators.volume_rendering")
smpl = importlib.import_module("3dhumangan_amd.lib.components.smpl")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="integrate,geo")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--R", type=int, default=4608)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--F", type=int, default=256)
    ap.add_argument("--engine", default="")
    a = ap.parse_args()
    dev = "cuda"
    res = {}
    if "integrate" in a.what:
        C = a.F + 3
        field = torch.randn(a.B, a.R, a.S, C + 1, device=dev)
        z = torch.sort(torch.rand(a.B, a.R, a.S, 1, device=dev) + 11, dim=2).values
        ms = timeit(lambda: vr.ray_integration(field, z, noise_std=0, clamp_mode="relu", last_back=True, white_back=True))
        by = 4 * a.B * a.R * (a.S * (C + 1) + a.S + C + 1 + a.S)
        res["ray_integrate"] = dict(ms=ms, GBps=by / ms / 1e6, bytes=by)
    if "geo" in a.what:
        cond = {k: v.to(dev) for k, v in synthetic.make_conditions(a.B, 6890, seed=0).items()}
        N = a.R * a.S
        pts = (torch.rand(a.B, N, 3, device=dev) - 0.5) * 2
        vik = smpl.vertex_inverse_transforms(cond["fk_matrices"], cond["lbs_weights"])
        ms = timeit(lambda: smpl.get_geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"],
                                                  cond["fk_matrices"], cond["lbs_weights"], vertex_ik=vik), iters=3, warmup=1)
        res["geo_features"] = dict(ms=ms, pairs_per_s=a.B * N * 6890 / ms * 1e3)
    if "field" in a.what or "fused" in a.what:
        impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
        H = a.F
        net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=H, hidden_dim=H, geo_feature_dim=31, output_dim=H + 4,
                                    feature_dim=H, num_blocks=4).to(dev).eval()
        if a.engine:
            net.precision = a.engine
        N = a.R * a.S
        pts = torch.rand(a.B, N, 3, device=dev) * 2 - 1
        geo = torch.rand(a.B, N, 31, device=dev) * 2 - 1
        fr = torch.randn(a.B, 4 * H, device=dev) * 0.5
        ph = torch.randn(a.B, 4 * H, device=dev)
        z = torch.sort(torch.rand(a.B, a.R, a.S, 1, device=dev) + 11, dim=2).values
        flop = 2 * (7 * H * H + 41 * H) * a.B * N
        if "field" in a.what:
            ms = timeit(lambda: net(pts, fr, ph, geo, None, input_scaler=0.7), iters=3, warmup=1)
            res["neural_field"] = dict(ms=ms, TFLOPs=flop / ms / 1e9)
        if "fused" in a.what:
            ms = timeit(lambda: net.render(pts, fr, ph, geo, None, z, a.S, input_scaler=0.7, last_back=True, white_back=True),
                        iters=3, warmup=1)
            res["render_fused"] = dict(ms=ms, TFLOPs=flop / ms / 1e9)
    if "synth" in a.what:
        configs = importlib.import_module("3dhumangan_amd.configs")
        gens = importlib.import_module("3dhumangan_amd.lib.generators")
        impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
        for nb, mod in ((9, [0, 1, 2]), (1, []), (2, []), (3, []), (1, [0])):
            cfg = {k: v for k, v in configs.MAP3DBN512.items() if isinstance(k, str)}
            cfg.update(gen_height=512, gen_width=512, synthesis_blocks=nb, mod_blocks=mod, dataset_length=2)
            cfg["neural_field_cls"] = impl.COORDCONCATSIREN
            G = gens.Map3DGenerator(**cfg).to(dev).eval()
            G.set_device(dev)
            if a.engine:
                G.synthesis_plan(dev).engine = a.engine
            fmap = torch.randn(a.B, 96 * 96, 256, device=dev)
            st = torch.randn(a.B, 1, 256, device=dev)
            ms = timeit(lambda: G._synthesize(fmap, st, (96, 96)), iters=3, warmup=1)
            stages = sum(32 if k in mod else 16 for k in range(nb)) * 2
            res[f"synth_nb{nb}_mod{len(mod)}"] = dict(ms=ms, stages=stages, us_per_stage_tile=ms * 1e3 / stages)
            del G
    print(json.dumps(res))


if __name__ == "__main__":
    main()
