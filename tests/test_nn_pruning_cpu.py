"""CPU model of the chunk-pruning rule of the nearest-vertex search (csrc/geo_features.hip, SORTED instantiations; reference
semantics: pytorch3d.ops.knn_points K = 1 at lib/components/smpl.py:220 -- squared L2, the smallest index among exact ties).

The kernel scans a Morton-sorted mesh chunk by chunk (64 vertices), keeps a running minimum `run` of an APPROXIMATE distance
(matrix-core filter, error <= e), remembers chunks within `tol` of it and refines those exactly.  Round 4 skips a chunk when its
bounding sphere lies outside every search sphere of the wave's points.  This test restates that rule in numpy (float32, the same
inflation constants) with an adversarial filter error, meshes with duplicated and mirror-symmetric vertices (exact distance ties)
and checks the winner against the brute force over the UNSORTED mesh, bit for bit; it also checks that the rule actually prunes."""
import numpy as np
import pytest

F = np.float32
CHUNK = 64


def morton_order(v):
    lo, hi = v.min(0), v.max(0)
    inv = np.where(hi > lo, F(1023.0) / (hi - lo), F(0)).astype(F)
    q = np.clip((v - lo) * inv, 0, 1023).astype(np.uint64)
    code = np.zeros(len(v), np.uint64)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    key = (code << np.uint64(32)) | np.arange(len(v), dtype=np.uint64)
    return np.argsort(key, kind="stable")


def chunk_spheres(vs):
    n = (len(vs) + CHUNK - 1) // CHUNK
    cen, rad = np.zeros((n, 3), F), np.zeros(n, F)
    for c in range(n):
        vv = vs[c * CHUNK:(c + 1) * CHUNK]
        cen[c] = F(0.5) * (vv.min(0) + vv.max(0))
        d = vv - cen[c]
        r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).max()
        rad[c] = np.sqrt(F(r2)) * F(1.00001) + F(1.0e-7)
    return cen, rad


def exact_d(p, v):
    d = p[None, :] - v
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]          # the oracle's order of operations, float32


def pruned_search(points, mesh, rng, wave=256):
    """-> (indices, fraction of chunk scans executed).  `points` [N,3], `mesh` [V,3] float32."""
    order = morton_order(mesh)
    vs, ids = mesh[order], order.astype(np.int64)
    cen, rad = chunk_spheres(vs)
    n_ch = len(rad)
    v2max = F((mesh * mesh).sum(1).max())
    out = np.zeros(len(points), np.int64)
    scanned = total = 0
    for w0 in range(0, len(points), wave):
        P = points[w0:w0 + wave]
        p2 = (P * P).sum(1).astype(F)
        S = p2 + v2max
        tol = (F(3.0517578e-5) * S).astype(F)
        e = S * F(2.0 ** -17)
        sb = F(3) * tol + p2
        run = np.full(len(P), F(3.0e38))
        sk = np.full(len(P), F(3.0e18))
        cands = [[] for _ in P]
        cw = P.mean(0)
        seed = int(np.argmin(((cen - cw) ** 2).sum(1)))
        for c in [seed] + [c for c in range(n_ch) if c != seed]:
            total += 1
            d = P - cen[c]
            d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            reach = rad[c] + sk
            if not (d2 <= reach * reach).any():
                continue
            scanned += 1
            vv = vs[c * CHUNK:(c + 1) * CHUNK]
            a = (vv * vv).sum(1)[None, :] - F(2) * (P @ vv.T)                    # |v|^2 - 2 p.v
            a = a + rng.uniform(-1, 1, a.shape).astype(F) * e[:, None]           # adversarial filter error within the bound
            cm = a.min(1)
            hit = cm <= run + tol
            for k in np.nonzero(hit)[0]:
                cands[k].append((c, cm[k]))
                run[k] = min(run[k], cm[k])
                sk[k] = np.sqrt(max(run[k] + sb[k], F(0))) * F(1.00001)
        for k, p in enumerate(P):
            best = None
            for c, val in cands[k]:
                if val <= run[k] + tol[k]:
                    dd = exact_d(p, vs[c * CHUNK:(c + 1) * CHUNK])
                    for j, dj in enumerate(dd):
                        key = (dj, ids[c * CHUNK + j])
                        if best is None or key < best:
                            best = key
            out[w0 + k] = best[1]
    return out, scanned / total


def brute(points, mesh):
    return np.array([int(np.argmin(exact_d(p, mesh))) for p in points])          # argmin: the first index among exact minima


def body_like(rng, V):
    """points scattered around a few 'bones' (a crude body), unordered as in 3dhumangan_amd.synthetic.template_body"""
    a = rng.normal(0, 0.4, (12, 3))
    b = a + rng.normal(0, 0.3, (12, 3))
    k = rng.integers(0, 12, V)
    t = rng.random(V)[:, None]
    return (a[k] * (1 - t) + b[k] * t + rng.normal(0, 0.04, (V, 3))).astype(F)


@pytest.mark.parametrize("V,N,seed", [(700, 512, 0), (1500, 768, 1), (6890, 512, 2)])
def test_pruned_search_equals_brute_force(V, N, seed):
    rng = np.random.default_rng(seed)
    mesh = body_like(rng, V)
    # waves as the renderer makes them: 4 neighbouring rays x 64 jittered depth samples each (256 consecutive points)
    waves = []
    for _ in range(N // 256):
        o = rng.uniform(-1.0, 1.0, 2)
        rays = []
        for r in range(4):
            z = np.linspace(-0.5, 0.55, 64) + rng.uniform(-0.008, 0.008, 64)
            xy = np.tile(o + np.array([0.03 * r, 0.0]), (64, 1)) * (1 + 0.05 * z[:, None])
            rays.append(np.concatenate([xy, z[:, None]], 1))
        waves.append(np.concatenate(rays))
    pts = np.concatenate(waves).astype(F)
    got, frac = pruned_search(pts, mesh, rng)
    assert np.array_equal(got, brute(pts, mesh))
    print(f"V={V}: {frac:.3f} of the chunk scans executed")
    if V >= 6000:
        assert frac < 0.6, frac           # the rule prunes (the bench workload: ~0.2-0.35)


def test_exact_ties_go_to_the_smallest_original_index():
    """Duplicated vertices and a mirror-symmetric mesh: points on the symmetry plane are equidistant (bit for bit) from a vertex
    and its mirror image, which the Morton order separates and reverses."""
    rng = np.random.default_rng(5)
    half = body_like(rng, 600)
    half[:, 0] = np.abs(half[:, 0]) + F(0.01)
    mirror = half * np.array([-1, 1, 1], F)
    mesh = np.concatenate([mirror, half, half[:50]]).astype(F)        # mirror images FIRST (lower indices), then duplicates LAST
    perm = rng.permutation(len(mesh))
    mesh = mesh[perm]
    pts = rng.uniform(-1, 1, (512, 3)).astype(F)
    pts[:384, 0] = 0                                                   # on the symmetry plane: every nearest vertex is tied
    got, _ = pruned_search(pts, mesh, rng)
    ref = brute(pts, mesh)
    assert np.array_equal(got, ref)
    d = np.array([exact_d(p, mesh) for p in pts[:384]])
    assert ((d == d.min(1, keepdims=True)).sum(1) >= 2).all()          # the ties are real


def test_skip_rule_is_conservative():
    """Every vertex of a skipped chunk is farther than the running bound by more than the filter's error allows to matter:
    a = |p - v|^2 - |p|^2 > run + 2 tol for all of them (the inequality the exactness argument needs), in float64."""
    rng = np.random.default_rng(9)
    mesh = body_like(rng, 2000)
    order = morton_order(mesh)
    vs = mesh[order]
    cen, rad = chunk_spheres(vs)
    v2max = F((mesh * mesh).sum(1).max())
    for _ in range(2000):
        p = rng.uniform(-1.3, 1.3, 3).astype(F)
        p2 = F((p * p).sum())
        tol = F(3.0517578e-5) * (p2 + v2max)
        c = int(rng.integers(0, len(rad)))
        run = F(rng.uniform(-float(p2), 2.0))
        sk = np.sqrt(max(run + F(3) * tol + p2, F(0))) * F(1.00001)
        d = p - cen[c]
        d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
        reach = rad[c] + sk
        if d2 <= reach * reach:
            continue                                                   # not skipped: nothing to show
        vv = vs[c * CHUNK:(c + 1) * CHUNK].astype(np.float64)
        a = ((p.astype(np.float64)[None] - vv) ** 2).sum(1) - float(p2)
        assert (a > float(run) + 2 * float(tol)).all()
