"""Per-sample diagnosis of the one bad ray: fused-geo x2 render vs two-kernel x2 render on the ring-stress setup."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ring_stress as T  # noqa: E402

smpl = importlib.import_module("3dhumangan_amd.lib.components.smpl")
G, cfg, z, cond, jit = T._setup()
nf = G.neural_field
cap = {}
orig_geo, orig_two = nf.render_geo, nf.render


def geo_wrap(*a, **k):
    out = orig_geo(*a, **k)
    cap["geo"] = (a, k, out)
    return out


def two_wrap(*a, **k):
    out = orig_two(*a, **k)
    cap["two"] = (a, k, out)
    return out


nf.render_geo, nf.render = geo_wrap, two_wrap
G.fuse_geo = True
G.forward(z, cond, jitter=jit, **cfg)
G.fuse_geo = False
G.forward(z, cond, jitter=jit, **cfg)
(fg, dg, wg), (ft, dt_, wt) = cap["geo"][2], cap["two"][2]
b, r = 1, 238
print("feats diff on the ray (last 3 = rgb):", (fg[b, r] - ft[b, r]).abs().max().item(), fg[b, r, -3:].tolist(), ft[b, r, -3:].tolist())
print("feature-map part max diff:", (fg[b, r, :-3] - ft[b, r, :-3]).abs().max().item())
dw = (wg[b, r, :, 0] - wt[b, r, :, 0]).abs()
print("weights fused:", [round(v, 4) for v in wg[b, r, :, 0].tolist()])
print("weights two  :", [round(v, 4) for v in wt[b, r, :, 0].tolist()])
print("sum fused / two:", wg[b, r].sum().item(), wt[b, r].sum().item())
a = cap["geo"][0]
pts, idx = a[0], a[3]
S = 64
sl = slice(r * S, (r + 1) * S)
print("nn idx on the ray:", idx[b, sl].tolist())
geo_two = cap["two"][0][3]
print("geo feature range on the ray (two-kernel path): min", geo_two[b, sl].min().item(), "max", geo_two[b, sl].max().item())
print("any non-finite in geo:", (~torch.isfinite(geo_two[b, sl])).any().item())
print("points z range:", pts[b, sl, 2].min().item(), pts[b, sl, 2].max().item())
# neighbours for contrast
for rr in (237, 239):
    print("ray", rr, "rgb diff", (fg[b, rr, -3:] - ft[b, rr, -3:]).abs().max().item(), "weights diff", (wg[b, rr] - wt[b, rr]).abs().max().item())
# every ray: where do weights differ at all?
dall = (wg - wt).abs().amax(dim=(2, 3))
print("rays with weight diff > 1e-3:", torch.nonzero(dall > 1e-3).tolist()[:20])
drgb = (fg[..., -3:] - ft[..., -3:]).abs().amax(-1)
print("rays with rgb diff > 1e-3:", torch.nonzero(drgb > 1e-3).tolist()[:20])
dfm = (fg[..., :-3] - ft[..., :-3]).abs().amax(-1)
print("rays with feature-map diff > 1e-2:", torch.nonzero(dfm > 1e-2).tolist()[:20], "max", dfm.max().item())

print("---- E1: does the bad ray look like white_back = 1?")
ga, gk = cap["geo"][0], cap["geo"][1]
ta, tk = cap["two"][0], cap["two"][1]
print("kwargs:", {k: v for k, v in gk.items() if not torch.is_tensor(v)})
nf.render_geo, nf.render = orig_geo, orig_two
nf.precision = "f16x2"
f_two_white = nf.render(*ta, **{**tk, "white_back": True})[0]
f_geo_white = nf.render_geo(*ga, **{**gk, "white_back": True})[0]
f_geo = nf.render_geo(*ga, **gk)[0]
print("fused(white=0) - two(white=1) on the ray:", (f_geo[b, r] - f_two_white[b, r]).abs().max().item())
print("fused(white=1) - two(white=1) on the ray:", (f_geo_white[b, r] - f_two_white[b, r]).abs().max().item())
print("fused(white=1) vs two(white=1): rays off by > 1e-3:", torch.nonzero((f_geo_white - f_two_white).abs().amax(-1) > 1e-3).tolist()[:10])
# one item at a time, and a different number of rays
for sel in ([1], [0, 1], [1, 1, 1]):
    ii = torch.tensor(sel, device="cuda")
    a2 = tuple(t[ii] if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == 2 else t for t in ga)
    k2 = {k: (v[ii] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 else v) for k, v in gk.items()}
    t2 = tuple(t[ii] if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == 2 else t for t in ta)
    kt2 = {k: (v[ii] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 else v) for k, v in tk.items()}
    fa, fb = nf.render_geo(*a2, **k2)[0], nf.render(*t2, **kt2)[0]
    print("items", sel, "bad rays:", torch.nonzero((fa - fb).abs().amax(-1) > 1e-3).tolist()[:10])
nf.precision = "f16x3"
fa, fb = nf.render_geo(*ga, **gk)[0], nf.render(*ta, **tk)[0]
print("x3 engines: bad rays", torch.nonzero((fa - fb).abs().amax(-1) > 1e-3).tolist()[:10])

print("---- E2: flags")
nf.precision = "f16x2"
for lb in (True, False):
    for wb in (True, False):
        fa, da, wa = nf.render_geo(*ga, **{**gk, "last_back": lb, "white_back": wb})
        fb, db, wb_ = nf.render(*ta, **{**tk, "last_back": lb, "white_back": wb})
        bad = torch.nonzero((fa - fb).abs().amax(-1) > 1e-3).tolist()
        print(f"last_back={lb} white_back={wb}: bad rays {bad[:6]}; on (1,238): feats diff {(fa[1, 238] - fb[1, 238]).abs().max().item():.5f} "
              f"(min over channels {(fa[1, 238] - fb[1, 238]).min().item():.5f}) depth diff {(da[1, 238] - db[1, 238]).abs().max().item():.2e} "
              f"weights diff {(wa[1, 238] - wb_[1, 238]).abs().max().item():.2e}  w_last {wa[1, 238, -1, 0].item():.5f}")
