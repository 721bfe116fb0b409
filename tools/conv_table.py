"""The per-shape table of the convolution / weight-gradient kernels alone (bench.py's op_rooflines()['conv_x3_by_shape']).
usage: python tools/conv_table.py [forward|weight_gradient]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


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


want = sys.argv[1] if len(sys.argv) > 1 else None
for r in bench.conv_rooflines(timeit):
    if want is None or r["pass_"] == want:
        print(f'{r["shape"]:28s} {r["activations"]} {r["pass_"]:16s} {r["ms"]:7.3f} ms  pipe {r["mfma_pipe_util"]:.3f}  hbm {r["hbm_frac"]:.3f}')
