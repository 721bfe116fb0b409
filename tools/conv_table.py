"""GPU: the per-shape table of the convolution / weight-gradient kernels (bench.py: conv_rooflines) as text.
usage: python tools/conv_table.py [weight planes of the AMP tier: 1 | 2]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
if len(sys.argv) > 1:
    conv.AMP_WEIGHT_PLANES = int(sys.argv[1])


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print(f"AMP weight planes = {conv.AMP_WEIGHT_PLANES}")
print(f"{'kernel':18s} {'pass':22s} {'shape':28s} {'act':4s} {'ms':>8s} {'TFLOP/s':>9s} {'frac':>6s} {'pipe':>6s} {'GB/s':>8s}")
for r in bench.conv_rooflines(timeit):
    print(f"{r['kernel']:18s} {r['pass_']:22s} {r['shape']:28s} {r['activations']:4s} {r['ms']:8.4f} {r['achieved_TFLOPs']:9.1f} "
          f"{r['frac']:6.3f} {r['mfma_pipe_util']:6.3f} {r['hbm_GBs']:8.0f}")
