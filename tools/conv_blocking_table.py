"""Time h3d_conv_x3_ex at forced blockings NT = 8 / 4 / 2 for the discriminator's layer shapes (B = 4, config 4) -- the table behind
h3d_conv_x3_nt_for's rule.  usage: python tools/conv_blocking_table.py   (spawns itself with H3D_CONV_FILL=0 H3D_CONV_NT_MAX=n)"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4, 128, 128, 512, 256, 3), (4, 128, 128, 256, 128, 3), (4, 128, 256, 128, 64, 3), (4, 256, 256, 128, 64, 3), (4, 256, 256, 64, 32, 3),
          (4, 512, 256, 64, 32, 3), (4, 256, 512, 32, 16, 3), (4, 512, 512, 32, 16, 3), (4, 1024, 256, 32, 16, 3), (4, 512, 512, 16, 8, 3),
          (4, 512, 512, 8, 4, 3), (4, 256, 256, 64, 32, 1), (4, 512, 256, 32, 16, 1), (4, 1024, 256, 32, 16, 1)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
    g = torch.Generator().manual_seed(1)
    for (B, ci, co, H, W, k) in SHAPES:
        row = []
        for dt in (torch.float32, torch.float16):
            x = torch.randn(B, ci, H, W, generator=g).to("cuda", dt).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(co, ci, k, k, generator=g) * 0.03).to("cuda")
            for _ in range(3):
                conv._run_conv(x, w, None)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                conv._run_conv(x, w, None)
            e.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(e) / 30 * 1e3)
        print(f"{B}x{ci}->{co} {H}x{W} k{k}", *[f"{t:.1f}" for t in row])
    sys.exit(0)

res = {}
for nt in (8, 4, 2):
    env = dict(os.environ, H3D_CONV_FILL="0", H3D_CONV_NT_MAX=str(nt))
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        parts = line.split()
        if len(parts) >= 5 and parts[0][0].isdigit():
            res.setdefault(" ".join(parts[:3]), {})[nt] = (float(parts[3]), float(parts[4]))
lib = importlib.import_module("3dhumangan_amd._lib") if False else None
print("shape (B x Cin->Cout HxW k) | tiles | fp32 us at NT 8 / 4 / 2 | f16 us at NT 8 / 4 / 2")
for (B, ci, co, H, W, k) in SHAPES:
    key = f"{B}x{ci}->{co} {H}x{W} k{k}"
    r = res.get(key, {})
    tiles = (B * H * W + 127) // 128
    print(key, "|", tiles, "|", " / ".join(f"{r[n][0]:.1f}" if n in r else "-" for n in (8, 4, 2)), "|",
          " / ".join(f"{r[n][1]:.1f}" if n in r else "-" for n in (8, 4, 2)))
