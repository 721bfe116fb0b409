"""GPU: which 128-pixel tiles should the x2 monitor sample?  For full x2 and x3 images of bench.py's workload: the full-image
maximum of |x2 - x3| / max|rgb| per item against the maximum over (a) the strided sample the monitor takes now, (b) the K tiles
with the largest |rgb|, (c) the K tiles with the largest per-tile range (max - min), (d) K random tiles.
usage: python tools/monitor_sampling_study.py [seeds] [K]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

seeds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1234,1,2,7,8").split(",")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
G, cfg = bench.build_generator("MAP3DBN512", (512, 512), (96, 96), 64, dev)
plan = G.synthesis_plan(dev)
rows = []
for seed in seeds:
    z, cond, jitter = bench.make_inputs(cfg, 16, dev, seed=seed)
    keep = plan.x2_monitor_tol
    plan.x2_monitor_tol = 1e9
    G.neural_field.precision, plan.engine = "f16x2", "f16x2"
    x2 = G.forward(z, cond, jitter=jitter, **cfg)["rgbs"].double()
    plan.x2_monitor_tol = keep
    G.neural_field.precision, plan.engine = "f16x3", "bf16x3"
    x3 = G.forward(z, cond, jitter=jitter, **cfg)["rgbs"].double()
    G.neural_field.precision, plan.engine = "f16x2", "f16x2"
    B = x2.shape[0]
    den = x3.abs().amax(dim=(2, 3), keepdim=True)
    err = ((x2 - x3).abs() / den).amax(1).reshape(B, -1, 128)                  # [B, tiles, 128]: per pixel, worst channel
    terr = err.amax(-1)                                                       # per tile
    mag = (x2.abs() / den).amax(1).reshape(B, -1, 128)
    tmag = mag.amax(-1)
    trange = (x2 / den).amax(1).reshape(B, -1, 128).amax(-1) - (x2 / den).amin(1).reshape(B, -1, 128).amin(-1)
    nt = terr.shape[1]
    first, step = plan.monitor_tiles(512, 512)
    strided = torch.arange(first, nt, step, device=dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for b in range(B):
        full = float(terr[b].max())
        if full > 0.01:
            continue                                                          # a last-sample flip of the unrefined field: not the monitor's business
        s_stride = float(terr[b, strided].max())
        s_mag = float(terr[b, torch.topk(tmag[b], K).indices].max())
        s_rng = float(terr[b, torch.topk(trange[b], K).indices].max())
        s_rnd = float(terr[b, torch.randperm(nt, generator=g)[:K].to(dev)].max())
        half = torch.cat([torch.topk(tmag[b], K // 2).indices, strided[::2]])
        s_mix = float(terr[b, half].max())
        rows.append(dict(seed=seed, item=b, full=full, stride=s_stride, top_mag=s_mag, top_range=s_rng, random=s_rnd, mix=s_mix))
for key in ("stride", "top_mag", "top_range", "random", "mix"):
    r = sorted(x["full"] / x[key] for x in rows)
    print(key, "ratio full/sample: median %.2f  p90 %.2f  max %.2f" % (r[len(r) // 2], r[int(0.9 * len(r))], r[-1]))
print(json.dumps(rows[:8]))
