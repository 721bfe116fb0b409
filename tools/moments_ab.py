"""Where the BatchNorm moments of a SPADE's input cost least: h3d_channel_moments (a pass over the stored tensor) against
h3d_conv_x3_moments (the producing GEMM's epilogue), both followed by h3d_rows_sum_f64.  usage: python tools/moments_ab.py [rows] [width]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
spade = importlib.import_module("3dhumangan_amd.lib.components.ops.spade")

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 512 * 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
g = torch.Generator().manual_seed(1)
x = torch.randn(M, C, generator=g).to(dev)
w, b = (torch.randn(C, C, generator=g) * 0.06).to(dev), torch.randn(C, generator=g).to(dev)
k = spade.HipKernels()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


y = lin.gemm_x3(x, w, b)
y3 = y.view(4, M // 4, C)
_, partial = lin.gemm_x3(x, w, b, moments=True)
print(f"rows {M} width {C}")
print(f"gemm_x3                         {timed(lambda: lin.gemm_x3(x, w, b)):8.1f} us")
print(f"gemm_x3 + epilogue moments      {timed(lambda: lin.gemm_x3(x, w, b, moments=True)):8.1f} us")
print(f"channel_moments pass (+ row sum){timed(lambda: k.moments(y3)):8.1f} us")
print(f"row sum of the epilogue's rows  {timed(lambda: k._sum_rows(partial.unsqueeze(0))):8.1f} us  ({partial.shape[0]} rows)")
a, e = k.moments(y3), k._sum_rows(partial.unsqueeze(0))
print("relative difference of the two", float(((a - e).abs() / a.abs().clamp_min(1e-30)).max()))
conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
for (B, Cc, H, W) in ((4, 128, 256, 128), (4, 256, 128, 64), (4, 512, 64, 32), (4, 512, 32, 16)):
    xi = torch.randn(B, Cc, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wi = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.03).to(dev)
    t = timed(lambda: conv._run_conv(xi, wi, None))
    print(f"conv 3x3 fp32 B{B} C{Cc} {H}x{W}: {t:8.1f} us  {2 * B * H * W * Cc * Cc * 9 / t / 1e6:7.1f} TFLOP/s")
    xh = xi.half()
    t = timed(lambda: conv._run_conv(xh, wi, None))
    print(f"conv 3x3 f16  B{B} C{Cc} {H}x{W}: {t:8.1f} us  {2 * B * H * W * Cc * Cc * 9 / t / 1e6:7.1f} TFLOP/s")
xh = x.half()
print(f"gemm f16 0.5M x 256 x 256 (own)   {timed(lambda: lin.gemm_x3(xh, w, b)):8.1f} us")
wh, bh = w.half(), b.half()
print(f"gemm f16 0.5M x 256 x 256 (library){timed(lambda: torch.nn.functional.linear(xh, wh, bh)):8.1f} us")
