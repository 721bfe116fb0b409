"""The condition front-end of the generator, in plain torch -- no pytorch3d / smplx / dataset files:

  preprocess_smpl_fix_body   one SMPL prediction record -> the `conditions` entries the generator consumes
                             (reference: SHHQDataset._preprocess_smpl_fix_body, lib/data/datasets.py:117-181)
  CameraPreprocessor         the camera matrices for a requested view (reference: SHHQPreprocessor.forward /
                             forward_with_rotation / _forward_fix_body, lib/data/preprocessor.py:45-97)

Both are pinned to the reference's own methods by tests/golden/frontend.npz.  Not reproduced: the pytorch3d mesh rasteriser
of the preprocessor (:138-176; it only feeds the discriminator's conditioning and the app's side-by-side view) --
`rasterized_semantics` comes back as an all-zero map.  Third-party boundary: pytorch3d.transforms.euler_angles_to_matrix
(pinned 0.6.2, not installed): euler_xyz_to_matrix follows its published definition, Rx(a) @ Ry(b) @ Rz(c) for "XYZ".
"""
import math

import torch

FOV = math.pi * 12 / 180
FOCAL = 1.0 / math.tan(FOV / 2)                      # 9.5144: intrinsics[0,0] of every sample


def euler_xyz_to_matrix(euler):
    """euler [B,3] (radians) -> [B,3,3] = Rx(e0) @ Ry(e1) @ Rz(e2)."""
    a, b, c = euler[:, 0], euler[:, 1], euler[:, 2]
    one, zero = torch.ones_like(a), torch.zeros_like(a)
    rx = torch.stack([one, zero, zero, zero, a.cos(), -a.sin(), zero, a.sin(), a.cos()], -1).view(-1, 3, 3)
    ry = torch.stack([b.cos(), zero, b.sin(), zero, one, zero, -b.sin(), zero, b.cos()], -1).view(-1, 3, 3)
    rz = torch.stack([c.cos(), -c.sin(), zero, c.sin(), c.cos(), zero, zero, zero, one], -1).view(-1, 3, 3)
    return rx @ ry @ rz


def preprocess_smpl_fix_body(pred, joints_index, smpl_tpose_vertices, inference=False):
    """pred: one SMPL regression record (arrays with a leading batch dim of 1, as the dataset stores them): orig_cam [1,4]
    (sx, sy, tx, ty), joints [1,Jall,3], full_pose [1,24,3,3], tpose_vertices [1,V,3], fk_matrices [1,24,4,4], lbs_weights
    [V,24] (, betas [1,10]).  -> dict of float32 tensors: scales, skeletons_xyz [24,3], intrinsics [4,4], vertices [V,3],
    tpose_vertices [V,3] (the TEMPLATE mesh raised by 0.35 in y), full_pose, fk_matrices, lbs_weights, cano_matrices, R, T.

    "fix body": the body is put into a canonical frame -- root rotation undone, then turned upside-up by Rx(pi) -- and the
    camera carries the view."""
    f64 = lambda x: torch.as_tensor(x, dtype=torch.float64)
    sx, _, tx, ty = [float(v) for v in torch.as_tensor(pred["orig_cam"]).reshape(-1)[:4].float()]     # stored as float32
    sx = sx / 2.0
    skeleton = f64(pred["joints"])[0].float().double()[list(joints_index)]
    K = torch.diag(torch.tensor([FOCAL, FOCAL, 1.0, 1.0], dtype=torch.float64))
    T = torch.eye(4, dtype=torch.float64)
    T[0, 3], T[1, 3], T[2, 3] = tx, ty, FOCAL / sx
    pose = f64(pred["full_pose"])[0]
    cano = torch.eye(4, dtype=torch.float64)
    rx_pi = torch.tensor([[1.0, 0, 0], [0, math.cos(math.pi), -math.sin(math.pi)], [0, math.sin(math.pi), math.cos(math.pi)]],
                         dtype=torch.float64)
    cano[:3, :3] = rx_pi @ torch.linalg.inv(pose[0])
    fk = torch.einsum("ij,bjk->bik", cano, f64(pred["fk_matrices"])[0])
    lbs = f64(pred["lbs_weights"])
    per_vertex = torch.einsum("bi,ijk->bjk", lbs, fk)
    tpose_shaped = f64(pred["tpose_vertices"])[0]
    homo = torch.cat([tpose_shaped, torch.ones_like(tpose_shaped[:, :1])], dim=1)
    vertices = torch.einsum("bij,bj->bi", per_vertex, homo)[:, :3]
    sk_h = torch.cat([skeleton, torch.ones_like(skeleton[:, :1])], dim=1)
    skeleton = torch.einsum("ij,bj->bi", cano, sk_h)[:, :3]
    template = torch.as_tensor(smpl_tpose_vertices).float().clone()
    template[..., 1] += 0.35
    out = {"scales": torch.tensor(sx, dtype=torch.float32), "skeletons_xyz": skeleton.float(), "intrinsics": K.float(),
           "vertices": vertices.float(), "tpose_vertices": template, "full_pose": pose.float(), "fk_matrices": fk.float(),
           "lbs_weights": lbs.float(), "cano_matrices": cano.float(), "R": torch.eye(4), "T": T.float()}
    if inference:
        out["body_shape"] = f64(pred["betas"])[0].float()
    return out


class CameraPreprocessor:
    """coordinate_mode "fix_body": world2cam = R @ T @ [root_rotation @ Rx(pi - v) Ry(-h) Rz(-r)], cam2world its inverse.
    Same call surface as the reference's SHHQPreprocessor (forward / forward_with_rotation / to)."""

    def __init__(self, device="cpu", **kwargs):
        self.device = device
        if kwargs.get("coordinate_mode", "fix_body") != "fix_body":
            raise NotImplementedError("only coordinate_mode='fix_body' (every shipped config) is provided")

    def to(self, device):
        self.device = device
        return self

    @torch.no_grad()
    def forward(self, data, rotate=False, **kwargs):
        """Random view: h, v ~ N(mean, stddev) when `rotate`, else the means (preprocessor.py:45-55; draws on the CPU RNG
        like the reference)."""
        B = data["scales"].shape[0]
        h = torch.randn(B) * (kwargs["h_stddev"] if rotate else 0) + kwargs["h_mean"]
        v = torch.randn(B) * (kwargs["v_stddev"] if rotate else 0) + kwargs["v_mean"]
        return self.forward_with_rotation(data, h, v, torch.zeros_like(h), **kwargs)

    __call__ = forward

    @torch.no_grad()
    def forward_with_rotation(self, data, h_rotation, v_rotation, r_rotation, **kwargs):
        data = dict(data)
        B = data["scales"].shape[0]
        dev = data["scales"].device
        euler = torch.zeros([B, 3], device=dev)
        euler[:, 1] = -h_rotation.reshape(B).to(dev)
        euler[:, 0] = math.pi - v_rotation.reshape(B).to(dev)
        euler[:, 2] = -r_rotation.reshape(B).to(dev)
        R = data["full_pose"][:, 0] @ euler_xyz_to_matrix(euler)
        body = torch.zeros(B, 4, 4, device=dev)
        body[:, :3, :3] = R
        body[:, 3, 3] = 1.0
        world2cam = torch.bmm(torch.bmm(data["R"], data["T"]), body)
        data["cam2world_matrices"] = torch.inverse(world2cam.float())
        data["raster_rotation"] = torch.inverse(R)                  # R_raster of the reference (feeds its rasteriser)
        h, w = kwargs.get("gen_height", 1), kwargs.get("gen_width", 1)
        data["rasterized_semantics"] = torch.zeros(B, 3, h, w, device=dev)
        return data
