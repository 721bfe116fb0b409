"""CPU test of the host-side logic and of the data-movement model of h3d_wgrad_x3 (csrc/wgrad_x3.hip): the K-slice partition
covers every row exactly once for ragged row counts, and a numpy restatement of the kernel's data movement -- slab loading
with row / column clipping, fragment reads (lane l: column l & 31, rows 8 (l >> 5) + e), the 32x32x16 MFMA contraction, the
accumulator layout (register i of lane l: row 8 (i >> 2) + 4 (l >> 5) + (i & 3), column l & 31) and the per-slice partial
outputs -- reproduces dY^T X.  (The arithmetic itself runs on the GPU: tests/test_gpu_train_path.py::test_wgrad_x3_vs_fp64.)"""
import importlib

import numpy as np
import pytest

L = importlib.import_module("3dhumangan_amd._lib")
KS = 16


def tiles_for(c):
    return 4 if c > 128 else 2 if c > 64 else 1


def rows_per_wg(M, slices):
    per = (M + slices - 1) // slices
    return ((per + KS - 1) // KS) * KS


@pytest.mark.parametrize("M,Co,Ci", [(1, 32, 32), (17, 32, 36), (255, 64, 64), (256, 256, 256), (70001, 256, 128),
                                     (524288, 256, 256), (589824, 256, 512), (40000, 768, 256)])
def test_slices_partition_the_rows(M, Co, Ci):
    slices = L.load().h3d_wgrad_x3_slices(M, Co, Ci)
    assert 1 <= slices <= max(1, (M + 255) // 256)
    per = rows_per_wg(M, slices)
    assert per % KS == 0 and per * slices >= M                        # every row belongs to exactly one slice [s per, (s+1) per)
    empty = slices - (M + per - 1) // per                             # rounding per to the k-step can leave trailing slices empty
    assert 0 <= empty <= slices // 8 + 1                              # (they write zeros); never more than a few


def emulate(dY, X, slices):
    """The kernel's data movement in numpy (float64 arithmetic): returns partial [slices, Co, Ci]."""
    M, Co = dY.shape
    Ci = X.shape[1]
    NA, NB = tiles_for(Co), tiles_for(Ci)
    WA, WB = 64 * NA, 64 * NB
    per = rows_per_wg(M, slices)
    out = np.zeros((slices, Co, Ci))
    lanes = np.arange(64)
    for s in range(slices):
        r_begin, r_end = s * per, min((s + 1) * per, M)
        for by in range((Co + WA - 1) // WA):
            for bz in range((Ci + WB - 1) // WB):
                co0, ci0 = by * WA, bz * WB
                acc = np.zeros((4, NA, NB, 64, 16))                   # [wave][a][b][lane][register]
                for row0 in range(r_begin, max(r_end, r_begin), KS):
                    slabA, slabB = np.zeros((KS, WA)), np.zeros((KS, WB))
                    for slab, src, c0, nc in ((slabA, dY, co0, Co), (slabB, X, ci0, Ci)):
                        W4 = slab.shape[1] // 4
                        for idx in range(KS * W4):                    # one float4 per (thread, j): idx = j * 256 + t
                            row, c = idx // W4, (idx % W4) * 4
                            if row0 + row < r_end and c0 + c < nc:
                                slab[row, c:c + 4] = src[row0 + row, c0 + c:c0 + c + 4]
                    for wave in range(4):
                        wy, wx = wave >> 1, wave & 1
                        for a in range(NA):
                            # fragment: lane l, element e <-> slab row 8 (l >> 5) + e, column (wy NA + a) 32 + (l & 31)
                            fa = np.stack([slabA[8 * (lanes >> 5) + e, (wy * NA + a) * 32 + (lanes & 31)] for e in range(8)], 1)
                            for b in range(NB):
                                fb = np.stack([slabB[8 * (lanes >> 5) + e, (wx * NB + b) * 32 + (lanes & 31)] for e in range(8)], 1)
                                # MFMA 32x32x16: A[m][k], B[k][n] with m = n = lane & 31, k = 8 (lane >> 5) + e
                                Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
                                for l in range(64):
                                    Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = fa[l]
                                    Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = fb[l]
                                D = Am @ Bm
                                for l in range(64):
                                    for i in range(16):
                                        acc[wave, a, b, l, i] += D[8 * (i >> 2) + 4 * (l >> 5) + (i & 3), l & 31]
                for wave in range(4):
                    wy, wx = wave >> 1, wave & 1
                    for a in range(NA):
                        for b in range(NB):
                            for l in range(64):
                                ci = ci0 + (wx * NB + b) * 32 + (l & 31)
                                for i in range(16):
                                    co = co0 + (wy * NA + a) * 32 + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3)
                                    if co < Co and ci < Ci:
                                        out[s, co, ci] = acc[wave, a, b, l, i]
    return out


@pytest.mark.parametrize("M,Co,Ci,slices", [(45, 36, 40, 2), (70, 72, 32, 3), (33, 132, 68, 1)])
def test_data_movement_model_reproduces_the_product(M, Co, Ci, slices):
    rng = np.random.default_rng(M)
    dY, X = rng.standard_normal((M, Co)), rng.standard_normal((M, Ci))
    got = emulate(dY, X, slices).sum(0)
    assert np.allclose(got, dY.T @ X, rtol=1e-12, atol=1e-12)
