"""Diagnosis: x2 engines with / without A4 fused into the render vs the strict fp32 engines on the ring-stress setup."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err  # noqa: E402
import test_gpu_ring_stress as T  # noqa: E402

for last_back in (True, False):
    G, cfg, z, cond, jit = T._setup()
    cfg["last_back"] = last_back
    outs = {}
    for name, fuse, prec, eng in (("x2 fused geo", True, "f16x2", "f16x2"), ("x2 two-kernel geo", False, "f16x2", "f16x2"),
                                  ("x3 fused geo", True, "f16x3", "bf16x3"), ("f32 strict", False, "f32", "f32")):
        G.fuse_geo = fuse
        G.neural_field.precision = prec
        G.synthesis_plan("cuda").engine = eng
        o = G.forward(z, cond, jitter=jit, **cfg)
        outs[name] = (o["rgbs"].cpu(), o["rgbs_render"].cpu())
    ref = outs["f32 strict"]
    for name, (rgb, ren) in outs.items():
        d = (rgb - ref[0]).abs()
        print(f"last_back={last_back} {name:20s} rgbs err {rel_err(rgb, ref[0]):.3e}  render err {rel_err(ren, ref[1]):.3e}  "
              f"pixels off by > 1e-2: {int((d.amax(1) > 1e-2).sum())} of {d[:, 0].numel()}; max |rgb| {float(ref[0].abs().max()):.3f}")
    plan = G.synthesis_plan("cuda")
    print("   x2 guard fell back:", plan.x2_fell_back() if hasattr(plan, "x2_fell_back") else None)

print("---- which rays")
G, cfg, z, cond, jit = T._setup()
res = {}
for name, fuse, prec in (("fused", True, "f16x2"), ("two", False, "f16x2")):
    G.fuse_geo = fuse
    G.neural_field.precision = prec
    G.synthesis_plan("cuda").engine = "f16x2"
    res[name] = G.forward(z, cond, jitter=jit, **cfg)["rgbs_render"].cpu()
d = (res["fused"] - res["two"]).abs().amax(1)          # [B, 96, 96]
bad = torch.nonzero(d > 1e-3)
print("bad rays:", bad.shape[0], "of", d.numel())
for b, y, x in bad.tolist()[:40]:
    r = y * 96 + x
    print(f"  item {b} ray ({y},{x}) index {r} (r % 4 = {r % 4}, r // 4 = {r // 4}) err {float(d[b, y, x]):.3f}  fused {res['fused'][b, :, y, x].tolist()} two {res['two'][b, :, y, x].tolist()}")
