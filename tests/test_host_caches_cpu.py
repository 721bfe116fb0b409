"""Host-side helpers added in round 4 that need no GPU: the per-parameter-version cache of f16 casts (ops/linear.py), and the
signature test bench.py / the ring-stress test use for rays on the reference's last-sample discontinuity."""
import importlib
import os
import sys

import torch

lin = importlib.import_module("3dhumangan_amd.lib.components.ops.linear")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_half_cast_is_cached_per_parameter_version():
    w = torch.nn.Parameter(torch.randn(8, 8))
    a = lin._half_cached(w)
    assert a.dtype == torch.float16 and lin._half_cached(w) is a            # same object, same version: one cast
    with torch.no_grad():
        w.add_(1.0)                                                         # what an optimiser step does: the version moves on
    b = lin._half_cached(w)
    assert b is not a and torch.equal(b, w.detach().half())
    assert lin._half_cached(None) is None
    h = torch.randn(4).half()
    assert lin._half_cached(h) is h                                         # already f16: nothing to do


def test_half_cast_cache_does_not_confuse_tensors_that_reuse_an_id():
    seen = []
    for i in range(50):                                                     # temporaries: ids get recycled by the allocator
        t = torch.full((4,), float(i))
        seen.append((lin._half_cached(t), float(i)))
        del t
    for h, v in seen:
        assert torch.equal(h, torch.full((4,), v).half())


def test_discontinuity_signature():
    import bench
    dr = torch.zeros(1, 3, 6)
    dr[0, :, 1] = 0.9949                      # uniform shift by the remaining transmittance: the discontinuity
    dr[0, :, 2] = torch.tensor([0.9, 0.1, 0.5])   # a large error that differs per channel: a real error
    dr[0, :, 3] = 5e-4                        # within tolerance
    dr[0, :, 4] = -0.73                       # uniform negative shift (the flip in the other direction)
    dr[0, :, 5] = torch.tensor([0.02, 0.02, 0.0215])   # uniform to 1.5e-3 only: not the signature
    flip = bench.discontinuity_rays(dr)
    assert flip.tolist() == [[False, True, False, False, True, False]]
