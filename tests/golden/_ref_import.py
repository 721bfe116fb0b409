"""Import shim for the Python reference at /root/reference (BUILD CONTAINER ONLY).

The reference's generator path imports a handful of third-party packages that
are not installed here (smplx, pytorch3d, torchvision, cv2, imageio).  None of
them is *executed* on the generator hot path except ``pytorch3d.ops.knn_points``
/ ``knn_gather`` (pinned 0.6.2, call sites lib/components/smpl.py:220-233), for
which this module provides a functional stand-in that follows the published
contract: squared L2 distances, ascending, ``idx [B,P,K]``.  Parity at that
boundary is therefore *unpinned* by the reference (documented in DESIGN.md).

Nothing in here ships to the GPU box; it is used by ``make_golden.py`` to
produce the committed ``*.npz`` fixtures.
"""
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _knn_points(p1, p2, K=1, **_):
    # direct-difference squared distances, same summation order as the build's
    # oracle/HIP kernel: (dx*dx + dy*dy) + dz*dz; first index wins ties.
    out_d, out_i = [], []
    for b in range(p1.shape[0]):
        a = p1[b]
        v = p2[b]
        dists = []
        idxs = []
        for s in range(0, a.shape[0], 8192):
            c = a[s:s + 8192]
            dx = c[:, None, 0] - v[None, :, 0]
            dy = c[:, None, 1] - v[None, :, 1]
            dz = c[:, None, 2] - v[None, :, 2]
            d2 = (dx * dx + dy * dy) + dz * dz
            d, i = torch.topk(d2, K, dim=1, largest=False, sorted=True)
            if K == 1:
                # topk does not promise first-index on ties; argmin-style pick:
                m = d2.min(dim=1, keepdim=True).values
                big = torch.full_like(d2, d2.shape[1], dtype=torch.long)
                ar = torch.arange(d2.shape[1]).expand_as(d2)
                i = torch.where(d2 == m, ar, big).min(dim=1, keepdim=True).values
                d = m
            dists.append(d)
            idxs.append(i)
        out_d.append(torch.cat(dists))
        out_i.append(torch.cat(idxs))
    return torch.stack(out_d), torch.stack(out_i), None


def _knn_gather(x, idx):
    B, P, K = idx.shape
    C = x.shape[-1]
    return torch.gather(x[:, :, None, :].expand(B, x.shape[1], K, C), 1,
                        idx[..., None].expand(B, P, K, C))


def _euler_angles_to_matrix(euler_angles, convention):
    """Stand-in for pytorch3d.transforms.euler_angles_to_matrix (pinned 0.6.2; call sites lib/data/preprocessor.py:85,
    107, 114), following its published definition: the product of the per-axis rotations in the order of `convention`
    (intrinsic), e.g. "XYZ" -> Rx(a0) @ Ry(a1) @ Rz(a2).  Like the kNN stand-in this boundary is unpinned by the reference."""
    def axis(ax, a):
        c, s, one, zero = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
        rows = {"X": (one, zero, zero, zero, c, -s, zero, s, c), "Y": (c, zero, s, zero, one, zero, -s, zero, c),
                "Z": (c, -s, zero, s, c, zero, zero, zero, one)}[ax]
        return torch.stack(rows, -1).reshape(a.shape + (3, 3))
    mats = [axis(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return mats[0] @ mats[1] @ mats[2]


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True

    class _Stub:
        def __init__(self, *a, **k):
            pass

    class _SMPL(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def _nf(*a, **k):
        raise NotImplementedError("shimmed third-party function")

    smplx = _mod("smplx")
    smplx.body_models = _mod("smplx.body_models", SMPL=_SMPL)
    smplx.utils = _mod("smplx.utils", Tensor=torch.Tensor, SMPLOutput=_Stub)
    smplx.lbs = _mod("smplx.lbs", blend_shapes=_nf, vertices2joints=_nf,
                     batch_rodrigues=_nf, batch_rigid_transform=_nf)

    p3d = _mod("pytorch3d")
    p3d.ops = _mod("pytorch3d.ops", knn_points=_knn_points, knn_gather=_knn_gather)
    p3d.renderer = _mod("pytorch3d.renderer", PerspectiveCameras=_Stub,
                        MeshRasterizer=_Stub, RasterizationSettings=_Stub)
    p3d.structures = _mod("pytorch3d.structures", Meshes=_Stub)
    p3d.transforms = _mod("pytorch3d.transforms", euler_angles_to_matrix=_euler_angles_to_matrix)

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    tv.utils = _mod("torchvision.utils", make_grid=_nf, save_image=_nf)
    tv.models = _mod("torchvision.models")
    tv.datasets = _mod("torchvision.datasets")
    _mod("cv2")
    _mod("imageio", mimwrite=_nf, imwrite=_nf)

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
