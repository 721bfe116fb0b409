"""The one-pass adjoint of the bilinear resize (csrc/bilinear.hip: bilinear_cl_adjoint_axis) restated on the CPU: per output
column, walk the source positions once with two running sums (the source index of align_corners=False interpolation is monotone)
and write every finished row exactly once.  Checked against autograd through F.interpolate for up-scaling, down-scaling, a single
source row and equal sizes, per axis and composed (rows, then columns) as h3d_bilinear_resize_cl_bwd does."""
import numpy as np
import pytest
import torch


def src_index(dst, ratio, n_in):
    s = max((np.float32(dst) + np.float32(0.5)) * np.float32(ratio) - np.float32(0.5), np.float32(0.0))
    i0 = min(int(s), n_in - 1)
    return i0, min(i0 + 1, n_in - 1), float(np.float32(s) - np.float32(i0))


def adjoint_axis(src, n):
    """src [N, R] -> dst [n, R]: dst[i] = sum_I weight(I -> i) src[I], one pass over I."""
    N, R = src.shape
    ratio = np.float32(n) / np.float32(N)
    dst = np.full((n, R), np.nan)
    acc0, acc1, cur = np.zeros(R), np.zeros(R), 0
    for I in range(N):
        i0, i1, t = src_index(I, ratio, n)
        assert i0 >= cur, "the walker relies on a monotone source index"
        while cur < i0:
            dst[cur], acc0, acc1, cur = acc0, acc1, np.zeros(R), cur + 1
        acc0 = acc0 + (1.0 - t) * src[I]
        if i1 == i0:
            acc0 = acc0 + t * src[I]
        else:
            acc1 = acc1 + t * src[I]
    while cur < n:
        dst[cur], acc0, acc1, cur = acc0, acc1, np.zeros(R), cur + 1
    assert not np.isnan(dst).any(), "every row written exactly once"
    return dst


@pytest.mark.parametrize("h,w,H,W", [(6, 5, 20, 12), (12, 6, 64, 32), (5, 7, 5, 7), (16, 16, 8, 9), (1, 3, 7, 2), (9, 4, 50, 23), (40, 3, 7, 3)])
def test_walker_is_the_adjoint_of_the_resize(h, w, H, W):
    g = torch.Generator().manual_seed(h * W)
    C = 3
    dy = torch.randn(1, C, H, W, generator=g, dtype=torch.float64)
    x = torch.zeros(1, C, h, w, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.interpolate(x, (H, W), mode="bilinear")
    (ref,) = torch.autograd.grad(y, x, dy)
    d = dy[0].permute(1, 2, 0).numpy()                                  # [H, W, C] channels-last
    rows = adjoint_axis(d.reshape(H, W * C), h).reshape(h, W, C)        # rows:    [H][W*C] -> [h][W*C]
    out = np.stack([adjoint_axis(rows[i].reshape(W, C), w) for i in range(h)])      # columns: [W][C] -> [w][C] per row
    assert np.abs(out - ref[0].permute(1, 2, 0).numpy()).max() < 1e-5 * max(1.0, float(ref.abs().max()))
