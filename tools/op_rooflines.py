"""GPU: the stand-alone HBM-bound ops through the C ABI on preallocated outputs (bench.py: op_rooflines), as text."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.conv_rooflines = lambda timeit: []
for k, v in bench.op_rooflines().items():
    if isinstance(v, dict):
        print(f"{k:28s} {v['ms']:.4f} ms  {v['achieved']:7.0f} GB/s  frac {v['frac']:.3f}  {v.get('shape', '')}  {v.get('matches_wrapper', '')}")
