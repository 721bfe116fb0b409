"""GPU: error of the f16x2 field engine against the reference's vectors (tests/golden/field_h256.npz) next to f16x3, and
timing of the fused render at the bench geometry.  usage: python tools/x2_report.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err  # noqa: E402

impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
g = load_golden("field_h256")
H = 256
for eng in ("f16x3", "f16x2"):
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=H, hidden_dim=H, geo_feature_dim=31, output_dim=H + 4, feature_dim=H, num_blocks=4)
    net.load_state_dict({k[len("neural_field."):]: v.float() for k, v in g["state"].items() if k.startswith("neural_field.")})
    net.precision = eng
    net = net.cuda().eval()
    out = net(g["points"].cuda(), g["freq"].cuda(), g["phase"].cuda(), g["geo"].cuda(), g["dirs"].cuda(), input_scaler=2.0 / 2.85).cpu()
    e = {k: rel_err(out[..., s], g["out"][..., s]) for k, s in (("rgb", slice(0, 3)), ("feat", slice(3, 3 + H)), ("sigma", slice(3 + H, 4 + H)))}
    print(eng, "vs reference vectors:", {k: f"{v:.2e}" for k, v in e.items()})
    # fused render timing: B=16, 96x96 rays x 64
    B, R, S = 16, 96 * 96, 64
    gen = torch.Generator().manual_seed(0)
    pts = (torch.rand(B, R * S, 3, generator=gen) * 2 - 1).cuda()
    geo = (torch.rand(B, R * S, 31, generator=gen) * 2 - 1).cuda()
    fr, ph = (torch.randn(B, 4 * H, generator=gen) * 0.5).cuda(), torch.randn(B, 4 * H, generator=gen).cuda()
    z = torch.sort(torch.rand(B, R, S, 1, generator=gen) + 11, dim=2).values.cuda()
    for _ in range(3):
        r = net.render(pts, fr, ph, geo, None, z, S, input_scaler=0.7)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        r = net.render(pts, fr, ph, geo, None, z, S, input_scaler=0.7)
    b.record()
    torch.cuda.synchronize()
    print(eng, f"fused render B=16 96x96x64: {a.elapsed_time(b) / 20:.2f} ms")
    if eng == "f16x3":
        base = [t.clone() for t in r]
    else:
        print("   f16x2 vs f16x3 render outputs:", [f"{rel_err(x.cpu(), y.cpu()):.2e}" for x, y in zip(r, base)])
