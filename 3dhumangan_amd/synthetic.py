"""Procedural stand-in for the SHHQ/SMPL `conditions` dict (A0 schema).

The reference builds this dict from a licensed SMPL model + dataset files
(lib/data/datasets.py:117-181, lib/data/preprocessor.py:72-97) which cannot be
shipped.  This module produces tensors with the same keys, shapes, dtypes and
geometric conventions from a seeded procedural body: a 24-joint kinematic tree,
`n_vertices` points on capsules around the bones, dense LBS weights whose rows
sum to one, a random (or canonical) pose driven through forward kinematics, and
the weak-perspective camera of the fix_body coordinate mode.

Everything is computed on CPU in fp32 so the same bytes can be handed to the
oracle and to the HIP path.
"""
import math

import torch

PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)

_REST = (
    (0.00, 0.00, 0.00), (0.07, -0.09, 0.00), (-0.07, -0.09, 0.00), (0.00, 0.11, -0.01),
    (0.10, -0.47, 0.01), (-0.10, -0.47, 0.01), (0.00, 0.25, 0.00), (0.09, -0.87, -0.02),
    (-0.09, -0.87, -0.02), (0.00, 0.30, 0.01), (0.11, -0.93, 0.10), (-0.11, -0.93, 0.10),
    (0.00, 0.51, -0.02), (0.08, 0.42, -0.01), (-0.08, 0.42, -0.01), (0.00, 0.60, 0.02),
    (0.17, 0.44, -0.02), (-0.17, 0.44, -0.02), (0.43, 0.44, -0.03), (-0.43, 0.44, -0.03),
    (0.68, 0.44, -0.03), (-0.68, 0.44, -0.03), (0.76, 0.44, -0.03), (-0.76, 0.44, -0.03),
)
_RADIUS = (0.12, 0.08, 0.08, 0.12, 0.06, 0.06, 0.12, 0.045, 0.045, 0.12, 0.04, 0.04,
           0.05, 0.06, 0.06, 0.09, 0.05, 0.05, 0.04, 0.04, 0.035, 0.035, 0.03, 0.03)

FOCAL = 1.0 / math.tan(math.pi * 12 / 180 / 2)     # lib/data/datasets.py:119-120


def _axis_angle_to_matrix(aa):
    theta = aa.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    k = aa / theta
    K = torch.zeros(aa.shape[0], 3, 3)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s = torch.sin(theta)[..., None]
    c = torch.cos(theta)[..., None]
    return torch.eye(3)[None] + s * K + (1 - c) * (K @ K)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=torch.float32)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=torch.float32)


def template_body(n_vertices=6890, seed=0):
    """-> rest joints [24,3], rest vertices [V,3], lbs weights [V,24]."""
    g = torch.Generator().manual_seed(1000 + seed)
    J = torch.tensor(_REST, dtype=torch.float32)
    rad = torch.tensor(_RADIUS, dtype=torch.float32)
    par = torch.tensor([max(p, 0) for p in PARENTS])
    a, b = J[par], J                                        # bone: parent -> joint
    blen = (b - a).norm(dim=-1)
    blen[0] = 0.10
    share = blen * rad
    bone = torch.multinomial(share / share.sum(), n_vertices, replacement=True, generator=g)
    t = torch.rand(n_vertices, generator=g)
    ang = torch.rand(n_vertices, generator=g) * 2 * math.pi
    axis = b[bone] - a[bone]
    axis[bone == 0] = torch.tensor([0.0, 0.10, 0.0])
    axis_n = axis / axis.norm(dim=-1, keepdim=True)
    helper = torch.where(axis_n[:, 1:2].abs() < 0.9, torch.tensor([[0.0, 1.0, 0.0]]), torch.tensor([[1.0, 0.0, 0.0]]))
    u = torch.cross(axis_n, helper.expand_as(axis_n), dim=-1)
    u = u / u.norm(dim=-1, keepdim=True)
    w = torch.cross(axis_n, u, dim=-1)
    r = rad[bone] * (0.85 + 0.3 * torch.rand(n_vertices, generator=g))
    V = a[bone] + axis * t[:, None] + r[:, None] * (torch.cos(ang)[:, None] * u + torch.sin(ang)[:, None] * w)
    # dense LBS weights: softmax over the 4 nearest joints, zero elsewhere
    d2 = torch.cdist(V, J).square()
    near = torch.topk(d2, 4, dim=1, largest=False)
    wts = torch.zeros(n_vertices, 24)
    wts.scatter_(1, near.indices, torch.softmax(-near.values / 0.01, dim=1))
    return J, V.contiguous(), wts.contiguous()


def forward_kinematics(J, rot):
    """rot [24,3,3] local rotations -> (A [24,4,4] rest->posed transforms, posed joints [24,3])."""
    G = [None] * 24
    for j in range(24):
        T = torch.eye(4)
        T[:3, :3] = rot[j]
        T[:3, 3] = J[j] - (J[PARENTS[j]] if PARENTS[j] >= 0 else torch.zeros(3))
        G[j] = T if PARENTS[j] < 0 else G[PARENTS[j]] @ T
    G = torch.stack(G)
    posed = G[:, :3, 3].clone()
    A = G.clone()
    A[:, :3, 3] = G[:, :3, 3] - torch.einsum("jab,jb->ja", G[:, :3, :3], J)
    return A, posed


def make_conditions(batch, n_vertices=6890, seed=0, pose_scale=0.5, scale=0.8, h_angle=0.0, v_angle=0.0):
    """Synthetic `conditions` dict (CPU fp32).  pose_scale=0 gives the canonical pose.

    Keys/shapes follow the consumer at lib/generators/map3d_generator.py:388-395."""
    J, V, W = template_body(n_vertices, seed)
    flip = torch.eye(4)
    flip[:3, :3] = _rot_x(math.pi)                          # cano_rotation, datasets.py:141
    out = {k: [] for k in ("skeletons_xyz", "vertices", "fk_matrices", "cam2world_matrices")}
    g = torch.Generator().manual_seed(2000 + seed)
    for b in range(batch):
        aa = (torch.rand(24, 3, generator=g) - 0.5) * 2 * pose_scale
        aa[0] = 0
        A, posed = forward_kinematics(J, _axis_angle_to_matrix(aa) if pose_scale > 0 else torch.eye(3).repeat(24, 1, 1))
        A = flip[None] @ A
        VA = torch.einsum("vj,jab->vab", W, A)
        verts = torch.einsum("vab,vb->va", VA, torch.cat([V, torch.ones(n_vertices, 1)], 1))[:, :3]
        joints = posed @ flip[:3, :3].T
        hb = h_angle if isinstance(h_angle, float) else float(h_angle[b])
        vb = v_angle if isinstance(v_angle, float) else float(v_angle[b])
        body_rot = torch.eye(4)
        body_rot[:3, :3] = _rot_x(math.pi - vb) @ _rot_y(-hb)   # preprocessor.py:82-88 with identity root
        T = torch.eye(4)
        T[2, 3] = FOCAL / scale
        out["skeletons_xyz"].append(joints)
        out["vertices"].append(verts)
        out["fk_matrices"].append(A)
        out["cam2world_matrices"].append(torch.inverse(T @ body_rot))
    cond = {k: torch.stack(v).float().contiguous() for k, v in out.items()}
    tp = V.clone()
    tp[:, 1] += 0.35                                        # datasets.py:159-160
    cond["tpose_vertices"] = tp[None].repeat(batch, 1, 1).contiguous()
    cond["lbs_weights"] = W[None].repeat(batch, 1, 1).contiguous()
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = FOCAL
    cond["intrinsics"] = K[None].repeat(batch, 1, 1).contiguous()
    cond["scales"] = torch.full((batch,), float(scale))
    # camera pieces the reference's preprocessor consumes (datasets.py:127-139, preprocessor.py:80-93)
    T = torch.eye(4)
    T[2, 3] = FOCAL / scale
    cond["R"] = torch.eye(4)[None].repeat(batch, 1, 1)
    cond["T"] = T[None].repeat(batch, 1, 1)
    cond["full_pose"] = torch.eye(3)[None, None].repeat(batch, 24, 1, 1)
    return cond


# The camera front-end lives in lib/data/conditions.py (pinned to the reference's preprocessor); the old names stay importable.
from .lib.data.conditions import CameraPreprocessor as SyntheticPreprocessor, euler_xyz_to_matrix  # noqa: E402,F401
