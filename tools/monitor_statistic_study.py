"""GPU: which STATISTIC of the monitor's sample predicts the full-image maximum of |x2 - x3| / max|rgb| best?  For each of the
sample's (32 strided 128-pixel tiles) maximum, RMS, mean |.|, 99 % and 99.9 % quantiles: k = the largest full / statistic ratio
over the items (the calibrated factor), its spread (k / median ratio), and how many items the rule `k * statistic > budget` sends
to the x3 engine.  usage: python tools/monitor_statistic_study.py [seeds] [budget]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

seeds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1234,1,2,3,7,8").split(",")]
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 6e-4
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
    err = ((x2 - x3).abs() / den).reshape(B, 3, -1, 128)                       # [B, 3, tiles, 128]
    first, step = plan.monitor_tiles(512, 512)
    strided = torch.arange(first, err.shape[2], step, device=dev)
    for b in range(B):
        full = float(err[b].max())
        if full > 0.01:
            continue                                                          # a last-sample flip of the unrefined field
        s = err[b][:, strided].reshape(-1)
        q = torch.quantile(s, torch.tensor([0.99, 0.999], device=dev, dtype=s.dtype))
        rows.append(dict(seed=seed, item=b, full=full, max=float(s.max()), rms=float(s.pow(2).mean().sqrt()), mean=float(s.mean()),
                         p99=float(q[0]), p999=float(q[1]), l4=float(s.pow(4).mean().pow(0.25)), l8=float(s.pow(8).mean().pow(0.125))))
need = sum(r["full"] > budget for r in rows)
print(f"{len(rows)} items; {need} with a full-image error over the budget {budget:g}")
for key in ("max", "l8", "l4", "p999", "p99", "rms", "mean"):
    ratios = sorted(r["full"] / r[key] for r in rows)
    k = ratios[-1]
    flagged = sum(k * r[key] > budget for r in rows)
    print(f"{key:5s} k = {k:7.3f}  median {ratios[len(ratios) // 2]:7.3f}  spread {k / ratios[len(ratios) // 2]:.3f}  flagged {flagged}")
print(json.dumps(rows))
