"""The inference app's harness logic against frames produced by the REFERENCE's generate_frames (CPU)."""
import importlib
import math
import sys

import numpy as np
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from _stub_generator import StubGenerator, StubPreprocessor  # noqa: E402

app = importlib.import_module("3dhumangan_amd.apps.sample_from_generator")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")


def test_generate_frames_matches_reference_harness():
    g = load_golden("app_harness")
    cfg = dict(latent_dim=16, gen_height=12, gen_width=6)
    cond = {"dummy": torch.arange(6.0).view(1, 6)}
    for baf in (0, 1):
        G = StubGenerator()
        frames, sem = app.generate_frames(G, StubPreprocessor(), cfg, 7, cond, 5, 0.5, 0.2, bool(baf))
        assert frames.dtype == np.uint8 and frames.shape == (5, 12, 6, 3)
        assert torch.equal(torch.cat([c[0] for c in G.calls]), g[f"z{baf}"])            # seed -> z convention
        assert torch.allclose(torch.cat([c[1] for c in G.calls]), g[f"c2w{baf}"])        # angle schedule
        assert np.array_equal(frames, g[f"frames{baf}"].numpy())                          # clamp + uint8 + NHWC
        assert np.array_equal(sem, g[f"sem{baf}"].numpy())


def test_synthetic_preprocessor_camera_math():
    cond = synthetic.make_conditions(2, n_vertices=32, seed=1)
    pre = synthetic.SyntheticPreprocessor()
    zero = torch.zeros(2, 1)
    out = pre.forward_with_rotation(cond, zero, zero, zero, gen_height=4, gen_width=2)
    # zero rotation reproduces the canonical camera baked into make_conditions
    assert torch.allclose(out["cam2world_matrices"], cond["cam2world_matrices"], atol=2e-5)
    h = torch.full((2, 1), math.pi / 6)
    out = pre.forward_with_rotation(cond, h, zero, zero, gen_height=4, gen_width=2)
    want = synthetic.make_conditions(2, n_vertices=32, seed=1, h_angle=math.pi / 6)["cam2world_matrices"]
    assert torch.allclose(out["cam2world_matrices"], want, atol=2e-5)
    r = out["cam2world_matrices"][:, :3, :3]
    assert torch.allclose(r @ r.transpose(1, 2), torch.eye(3).expand(2, 3, 3), atol=1e-5)
    assert out["rasterized_semantics"].shape == (2, 3, 4, 2)
