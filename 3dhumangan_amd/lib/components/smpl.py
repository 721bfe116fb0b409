"""SMPL-conditioned geometry features (reference: lib/components/smpl.py:210-249), HIP-backed."""
import torch

import os

from ... import _lib

GEO_DIM = 31
# H3D_NN_PRUNE=0: the full scan of the unsorted mesh (h3d_geo_features / h3d_nearest_vertex) instead of the pruned scan of the
# Morton-sorted one (h3d_mesh_sort + h3d_*_sorted): the same indices bit for bit, for A/B measurements
PRUNE = os.environ.get("H3D_NN_PRUNE", "1") != "0"
TILED = os.environ.get("H3D_NN_TILED", "1") != "0"      # compact 8 x 8 x 4 boxes of samples per wave (nearest_vertex(ray_shape=...))


def sort_mesh(vertices):
    """vertices [B,V,3] fp32 (device) -> the workspace of h3d_mesh_sort: per pose the vertices in Morton order with their original
    indices, and the bounding spheres of the chunks of 64.  One launch, one workgroup per pose; None when the mesh is too large
    for the LDS sort (the callers then take the unsorted path)."""
    B, V, _ = vertices.shape
    lib = _lib.load()
    nbytes = lib.h3d_mesh_sort_bytes(B, V)
    vpad = (V + 63) // 64 * 64
    # the sorted search keeps FOUR planes of the mesh in LDS (x, y, z, original index): larger meshes take the full scan (three)
    if nbytes <= 0 or V > 16384 or vpad // 64 > 256 or 4 * (4 * vpad + 24 * 3 + 4) > 160 * 1024:
        return None
    ws = torch.empty(nbytes // 4, device=vertices.device, dtype=torch.float32)
    rc = lib.h3d_mesh_sort(_lib.ptr(vertices), _lib.ptr(ws), B, V, _lib.stream_handle())
    _lib.check(rc, "h3d_mesh_sort")
    return ws


def vertex_inverse_transforms(fk_matrices, lbs_weights):
    """[B,V,16]: per-vertex blended inverse bone transforms (smpl.py:217-218).  Once per pose; a
    [V,24]x[24,16] product per sample, done by the BLAS library on the device."""
    ik = torch.inverse(fk_matrices.float())
    B, J = ik.shape[:2]
    return torch.bmm(lbs_weights.float(), ik.reshape(B, J, 16)).contiguous()


def get_geo_features(points, skeletons, vertices, tpose_vertices, fk_matrices, lbs_weights, legacy_mode=False,
                     vertex_ik=None, return_index=False, out_stride=GEO_DIM):
    """Same arguments and result as the reference function: points [B,N,3] -> [B,N,31].

    ``vertex_ik`` lets a caller that evaluates many point sets for one pose reuse the blended transforms."""
    _lib.need_cuda(points, skeletons, vertices, tpose_vertices, fk_matrices, lbs_weights, vertex_ik)
    B, N, _ = points.shape
    V = vertices.shape[1]
    if vertex_ik is None:
        vertex_ik = vertex_inverse_transforms(fk_matrices, lbs_weights)
    # every converted tensor is bound to a local: ptr() keeps only the address, so a temporary would be freed (and its
    # block reused by the next same-sized temporary) before the kernel launches
    pts = points.contiguous().float()
    sk = skeletons.contiguous().float()
    vt = vertices.contiguous().float()
    tv = tpose_vertices.contiguous().float()
    vertex_ik = vertex_ik.contiguous().float()
    geo = torch.empty((B, N, out_stride), device=pts.device, dtype=torch.float32)
    if out_stride > GEO_DIM:
        geo[..., GEO_DIM:].zero_()
    idx = torch.empty((B, N), device=pts.device, dtype=torch.int32) if return_index else None
    ws = sort_mesh(vt) if PRUNE and B > 0 and N > 0 else None
    if ws is not None:
        rc = _lib.load().h3d_geo_features_sorted(_lib.ptr(pts), _lib.ptr(sk), _lib.ptr(ws), _lib.ptr(tv), _lib.ptr(vertex_ik),
                                                 _lib.ptr(geo), _lib.ptr(idx), B, N, V, out_stride, int(bool(legacy_mode)),
                                                 _lib.stream_handle())
    else:
        rc = _lib.load().h3d_geo_features(_lib.ptr(pts), _lib.ptr(sk), _lib.ptr(vt), _lib.ptr(tv), _lib.ptr(vertex_ik),
                                          _lib.ptr(geo), _lib.ptr(idx), B, N, V, out_stride, int(bool(legacy_mode)),
                                          _lib.stream_handle())
    _lib.check(rc, "h3d_geo_features")
    return (geo, idx) if return_index else geo


def nearest_vertex(points, vertices, ray_shape=None):
    """K = 1 nearest mesh vertex of every point (the search inside get_geo_features; pytorch3d.ops.knn_points at
    lib/components/smpl.py:220 of the reference): points [B,N,3], vertices [B,V,3] -> int32 [B,N].  Feeds the fused render
    kernels that build the geometry features themselves (COORDCONCATSIREN.render_geo).
    ray_shape = (Hr, Wr, S): the points are the samples of a render grid, [B, Hr, Wr, S, 3] flattened (round 5) -- the pruned
    search then works on compact 8 x 8 x 4 boxes of samples (h3d_nearest_vertex_sorted_rays; H3D_NN_TILED=0: the linear
    assignment); the same indices either way."""
    _lib.need_cuda(points, vertices)
    B, N, _ = points.shape
    pts = points.contiguous().float()
    vt = vertices.contiguous().float()
    idx = torch.empty((B, N), device=pts.device, dtype=torch.int32)
    ws = sort_mesh(vt) if PRUNE and B > 0 and N > 0 else None
    if ws is not None and ray_shape is not None and TILED and int(ray_shape[0]) * int(ray_shape[1]) * int(ray_shape[2]) == N:
        rc = _lib.load().h3d_nearest_vertex_sorted_rays(_lib.ptr(pts), _lib.ptr(ws), _lib.ptr(idx), B, int(ray_shape[0]),
                                                        int(ray_shape[1]), int(ray_shape[2]), vt.shape[1], _lib.stream_handle())
    elif ws is not None:
        rc = _lib.load().h3d_nearest_vertex_sorted(_lib.ptr(pts), _lib.ptr(ws), _lib.ptr(idx), B, N, vt.shape[1], _lib.stream_handle())
    else:
        rc = _lib.load().h3d_nearest_vertex(_lib.ptr(pts), _lib.ptr(vt), _lib.ptr(idx), B, N, vt.shape[1], _lib.stream_handle())
    _lib.check(rc, "h3d_nearest_vertex")
    return idx
