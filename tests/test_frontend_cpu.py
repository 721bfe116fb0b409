"""The condition front-end against the reference's own methods: SHHQDataset._preprocess_smpl_fix_body and
SHHQPreprocessor._forward_fix_body were run on a synthetic SMPL record in the build container
(tests/golden/make_golden_train.py -> frontend.npz)."""
import importlib
import math

import torch

from conftest import load_golden, rel_err

data_mod = importlib.import_module("3dhumangan_amd.lib.data")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")


def test_smpl_record_to_conditions():
    g = load_golden("frontend")
    pred = {k: v.numpy() for k, v in g["pred"].items()}
    out = data_mod.preprocess_smpl_fix_body(pred, g["joints_index"].tolist(), g["smpl_tpose_vertices"].numpy(), inference=True)
    assert set(out) == set(g["conditions"])
    for k, want in g["conditions"].items():
        got = out[k]
        assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape), k
        assert rel_err(got, want) < 2e-6, k
    assert abs(float(out["intrinsics"][0, 0]) - 1 / math.tan(math.pi * 6 / 180)) < 1e-6


def test_camera_matrices_for_requested_views():
    g = load_golden("frontend")
    B = g["cam2world"].shape[0]
    data = {k: v.float()[None].repeat(B, *([1] * v.dim())) for k, v in g["conditions"].items()}
    data["scales"] = data["scales"].reshape(B)
    pre = data_mod.CameraPreprocessor()
    out = pre.forward_with_rotation(data, g["angles"]["h"], g["angles"]["v"], g["angles"]["r"], gen_height=8, gen_width=4)
    assert rel_err(out["cam2world_matrices"], g["cam2world"]) < 2e-6
    assert rel_err(out["raster_rotation"], g["R_raster"]) < 2e-6
    assert out["rasterized_semantics"].shape == (B, 3, 8, 4) and "cam2world_matrices" not in data        # input dict untouched
    # forward(): the mean view without rotation noise, one per batch item
    o2 = pre.forward(data, rotate=False, h_stddev=0.4, v_stddev=0.1, h_mean=0.0, v_mean=0.0, gen_height=8, gen_width=4)
    zero = torch.zeros(B)
    assert torch.equal(o2["cam2world_matrices"], pre.forward_with_rotation(data, zero, zero, zero)["cam2world_matrices"])


def test_old_import_path_still_works():
    assert synthetic.SyntheticPreprocessor is data_mod.CameraPreprocessor
    e = torch.tensor([[0.3, -0.2, 0.1]])
    m = synthetic.euler_xyz_to_matrix(e)[0]
    assert torch.allclose(m @ m.t(), torch.eye(3), atol=1e-6) and abs(float(torch.det(m)) - 1) < 1e-6
