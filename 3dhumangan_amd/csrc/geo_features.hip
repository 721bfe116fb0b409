// A4 SMPL geometry features for gfx950.
// Reference semantics: lib/components/smpl.py:210-249 (get_geo_features); K=1 nearest vertex per
// pytorch3d.ops.knn_points contract (squared L2, first index wins exact ties).
//
// Workgroup = 512 threads, 4 points per thread.  The pose's mesh (V x 3 fp32 = 83 KB for SMPL) is staged once per
// workgroup into LDS as four SoA planes (x, y, z, |v|^2); every lane sweeps all V vertices, 4 per step, reading the
// planes with wave-uniform (broadcast) ds_read_b128.
//
// The search is filter + refine, and bit-exact:
//   filter  a(v) = |v|^2 - 2 p.v  (= |p - v|^2 - |p|^2) on the matrix cores: D[v][p] = A[v][k] * B[k][p] with
//           A = (vx, vy, vz, |v|^2), B = (-2px, -2py, -2pz, 1), both split into f16 hi + lo, and the three partial
//           products hi*hi, hi*lo, lo*hi laid side by side along K (12 of the 16 k-slots of one
//           v_mfma_f32_32x32x16_f16): ONE MFMA evaluates 32 vertices x 32 points to 22 significant bits per operand.
//           A wave filters its 256 points (eight B fragments, resident) against a 32-vertex tile with 8 MFMAs; only
//           the minimum of a per chunk of 64 consecutive vertices is kept (v_min3 over the accumulator registers).
//   refine  the oracle's own arithmetic, d = (dx*dx + dy*dy) + dz*dz with no fused multiply-add and a strict "<" in
//           ascending vertex order, runs only over candidate chunks.
// A chunk is a candidate when its filter minimum is within tol of the running filter minimum at the time it is
// scanned.  With e >= |a(v) + |p|^2 - d_float(v)| for every vertex (rounding of both formulas; e <= 28 eps S,
// S = |p|^2 + max|v|^2; here e <= 2^-17 S) and tol = 2^-15 S >= 2e: the exact winner w has a(w) <= d_w - |p|^2 + e, every vertex has
// a >= d_w - |p|^2 - e, so the chunk of w is within 2e of the final -- hence of the running -- minimum.  The last
// eight candidates are remembered (id + filter minimum) and those still within tol of the FINAL minimum are
// refined in scan order; a point that overflows the list (pathological ties) is refined over the whole mesh.
// The per-point tail gathers the blended inverse bone transform of the winner (64 B), canonicalises the point,
// gathers the T-pose vertex and evaluates the 24 joint distances.
//
// Round 4: chunk pruning on a spatially sorted mesh (SORTED instantiations, h3d_mesh_sort + h3d_*_sorted).  h3d_mesh_sort puts the
// pose's vertices in Morton order (one workgroup per pose, bitonic sort in LDS; key = 30-bit code << 32 | original index, so the
// order is a deterministic function of the mesh) and records a bounding sphere per chunk of 64.  A wave skips a chunk when, for
// every one of its 256 points, the sphere lies outside the point's current search radius:
//     |p - c| > r (1 + 1e-5) + s,    s^2 >= run + 3 tol + |p|^2      (run = the point's running filter minimum)
// so that every vertex of the chunk has a = |p - v|^2 - |p|^2 > run + 2 tol, hence a filter value > run + tol: the chunk
// would not have been a candidate and would not have lowered `run` -- the candidate lists, and with them the result, are those
// of the full scan of the same vertex order, bit for bit.  Exact ties are broken by the ORIGINAL vertex index (carried as a
// fourth LDS plane), which is what "first index wins" means for the unsorted mesh.  The chunk nearest to the wave's centroid is
// scanned first, so `run` is tight from the start: measured on the bench workload a wave scans ~22 % of the chunks
// (tests/test_nn_pruning_cpu.py models the rule; the GPU tests compare indices with the oracle's brute force).
#include "common.hpp"

namespace {

constexpr int kThreads = 512;
constexpr int kPts = 4;
constexpr int kJoints = 24;
constexpr int kChunk = 64;                 // vertices per filter chunk
constexpr int kCand = 6;                   // remembered candidate chunks per point (8 until round 4: the sorted kernels need the registers)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// two fp32 -> packed f16 hi halves (returned) and packed f16 lo halves (residuals)
__device__ __forceinline__ unsigned split2_f16(float a, float b, unsigned& lo) {
    const half2v h2 = __builtin_convertvector(f2{a, b}, half2v);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a - (float)h2.x, b - (float)h2.y}, half2v));
    return __builtin_bit_cast(unsigned, h2);
}

__device__ __forceinline__ float sqdist_exact(float px, float py, float pz, float vx, float vy, float vz) {
    const float dx = __fsub_rn(px, vx), dy = __fsub_rn(py, vy), dz = __fsub_rn(pz, vz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// exact scan of vertices [v4_begin*4, v4_end*4): the smallest distance, and among exact minima the smallest vertex index.
// Unsorted mesh (ids == null): position = index, ascending scan, strict "<".  Sorted mesh: ids holds the original index of
// every position and the tie rule is explicit (the scan order is no longer the index order).
template <bool SORTED>
__device__ __forceinline__ void refine(const f4* vx4, const f4* vy4, const f4* vz4, const i4* ids4, int v4_begin, int v4_end,
                                       float px, float py, float pz, float& best, int& bi) {
#pragma clang loop unroll(disable)
    for (int v4 = v4_begin; v4 < v4_end; ++v4) {
        const f4 X = vx4[v4], Y = vy4[v4], Z = vz4[v4];
        const float d0 = sqdist_exact(px, py, pz, X.x, Y.x, Z.x);
        const float d1 = sqdist_exact(px, py, pz, X.y, Y.y, Z.y);
        const float d2 = sqdist_exact(px, py, pz, X.z, Y.z, Z.z);
        const float d3 = sqdist_exact(px, py, pz, X.w, Y.w, Z.w);
        if constexpr (SORTED) {
            // one 64-bit key per vertex: distance bits (non-negative floats order like their bit patterns) above the original index
            const i4 I = ids4[v4];
            unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi;
            const unsigned long long k0 = ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned)I.x;
            const unsigned long long k1 = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned)I.y;
            const unsigned long long k2 = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)I.z;
            const unsigned long long k3 = ((unsigned long long)__float_as_uint(d3) << 32) | (unsigned)I.w;
            key = k0 < key ? k0 : key;
            key = k1 < key ? k1 : key;
            key = k2 < key ? k2 : key;
            key = k3 < key ? k3 : key;
            best = __uint_as_float((unsigned)(key >> 32));
            bi = (int)(unsigned)key;
        } else {
            if (d0 < best) { best = d0; bi = v4 * 4 + 0; }
            if (d1 < best) { best = d1; bi = v4 * 4 + 1; }
            if (d2 < best) { best = d2; bi = v4 * 4 + 2; }
            if (d3 < best) { best = d3; bi = v4 * 4 + 3; }
        }
    }
}

// FEATURES = false: the search only -- nn_index is the whole output (h3d_nearest_vertex; the features are then built in the field
// kernel's prologue, csrc/field_x3.hip GEOIN).
// SORTED: `vertices` is the workspace of h3d_mesh_sort for this batch: per pose Vpad float4 (x, y, z, original index as bits; the
// padding positions at 3e18) followed by Vpad / 64 float4 chunk spheres (centre, inflated radius).
template <bool FEATURES, bool SORTED>
__global__ __launch_bounds__(kThreads) void geo_features_kernel(
    const float* __restrict__ points, const float* __restrict__ joints, const float* __restrict__ vertices,
    const float* __restrict__ tpose, const float* __restrict__ vertex_ik, float* __restrict__ geo,
    int32_t* __restrict__ nn_index, int64_t N, int V, int Vpad, int geo_stride, int legacy_mode, int tile_S, int tile_Wr,
    int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* vx = smem;
    float* vy = smem + Vpad;
    float* vz = smem + 2 * Vpad;
    int* vid = reinterpret_cast<int*>(smem + 3 * Vpad);          // SORTED only
    float* jl = smem + (SORTED ? 4 : 3) * Vpad;   // 24*3 joints
    unsigned* v2max_bits = reinterpret_cast<unsigned*>(jl + kJoints * 3);
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const float* __restrict__ vb = vertices + (int64_t)b * V * 3;
    const f4* __restrict__ ws = reinterpret_cast<const f4*>(vertices) + (int64_t)b * (Vpad + Vpad / kChunk);
    const f4* __restrict__ bounds = ws + Vpad;
    if (t == 0) *v2max_bits = 0u;
    __syncthreads();
    float v2 = 0.f;
    for (int i = t; i < Vpad; i += kThreads) {
        const bool ok = SORTED ? true : i < V;
        float x, y, z;
        if constexpr (SORTED) {
            const f4 q = ws[i];
            x = q.x; y = q.y; z = q.z;
            vid[i] = __float_as_int(q.w);
        } else {
            // padding vertices sit at +inf distance for the exact scan and can never win (the filter masks them itself)
            x = ok ? vb[i * 3 + 0] : 3.0e18f; y = ok ? vb[i * 3 + 1] : 3.0e18f; z = ok ? vb[i * 3 + 2] : 3.0e18f;
        }
        vx[i] = x; vy[i] = y; vz[i] = z;
        if (x < 1.0e18f) v2 = fmaxf(v2, x * x + y * y + z * z);
    }
    atomicMax(v2max_bits, __float_as_uint(v2));      // non-negative floats order like their bit patterns
    if (FEATURES && t < kJoints * 3) jl[t] = joints[(int64_t)b * kJoints * 3 + t];
    __syncthreads();
    const float v2max = __uint_as_float(*v2max_bits);

    // A wave owns 256 points as 8 sets of 32: lane (m, hh) owns the wave's points i = (4*hh + k) * 32 + m, k = 0..3.
    // Linear (tile_S == 0): point wbase + i -- 256 CONSECUTIVE points, i.e. 256 / S whole neighbouring rays: a thin slab as long as
    // the rays.  Tiled (round 5; the points are [Hr, Wr, S] rays x samples): an 8 x 8 patch of rays x 4 consecutive samples -- a
    // compact box, so that far fewer mesh chunks intersect the search spheres of ALL of the wave's points (the pruning rule is
    // per wave) -- wave w of a pose = (patch row, patch column, sample block); the result is the exact arg-min either way.
    const int lane = t & 63, m = lane & 31, hh = lane >> 5;
    const int64_t wave_id = (int64_t)blockIdx.x * (kThreads / 64) + (t >> 6);
    const int64_t wbase = wave_id * 256;
    int64_t tile_base = 0;
    bool tile_ok = true;
    if (tile_S > 0) {
        const int tiles_s = tile_S / 4;
        const int sz = (int)(wave_id % tiles_s);
        const int64_t pxy = wave_id / tiles_s;
        const int px_ = (int)(pxy % tiles_x), py_ = (int)(pxy / tiles_x);
        tile_ok = py_ < tiles_y;
        tile_base = (((int64_t)py_ * 8) * tile_Wr + px_ * 8) * tile_S + sz * 4;         // point index of (ray 0 of the patch, sample 4 sz)
    }
    auto point_of = [&](int i) -> int64_t {
        if (tile_S == 0) return wbase + i;
        if (!tile_ok) return N;                                                          // past the last patch row: masked like n >= N
        const int ri = i >> 2;                                                           // ray of the patch: (ri >> 3, ri & 7)
        return tile_base + ((int64_t)(ri >> 3) * tile_Wr + (ri & 7)) * tile_S + (i & 3);
    };
    float px[kPts], py[kPts], pz[kPts], best[kPts];
    int bi[kPts];
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = point_of((4 * hh + k) * 32 + m);
        const bool ok = n < N;
        const float* p = points + ((int64_t)b * N + (ok ? n : 0)) * 3;
        px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
        best[k] = 3.4e38f;
        bi[k] = 0;
    }
    const f4* vx4 = reinterpret_cast<const f4*>(vx);
    const f4* vy4 = reinterpret_cast<const f4*>(vy);
    const f4* vz4 = reinterpret_cast<const f4*>(vz);
    const i4* vid4 = reinterpret_cast<const i4*>(vid);

    // ---- filter operands: B fragments of the 8 point sets.  Lane (m, hh) holds k-slots 8*hh .. 8*hh+7 of column m:
    //      hh = 0: [B_hi | B_lo], hh = 1: [B_hi | 0]  against  A: hh = 0: [A_hi | A_hi], hh = 1: [A_lo | 0].
    half8 bfrag[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = j & 3;
        const float ox = __shfl_xor(px[k], 32, 64), oy = __shfl_xor(py[k], 32, 64), oz = __shfl_xor(pz[k], 32, 64);
        const bool own = (j >> 2) == hh;
        const float x = -2.f * (own ? px[k] : ox), y = -2.f * (own ? py[k] : oy), z = -2.f * (own ? pz[k] : oz);
        unsigned lo01, lo23;
        const unsigned hi01 = split2_f16(x, y, lo01), hi23 = split2_f16(z, 1.f, lo23);
        bfrag[j] = __builtin_bit_cast(half8, u4{hi01, hi23, hh ? 0u : lo01, hh ? 0u : lo23});
    }
    float run[kPts], tol[kPts], cval[kPts][kCand];
    float sk[kPts], sb[kPts];                 // SORTED: search radius of the skip test; 3 tol + |p|^2
    unsigned long long cid[kPts];
    int ncand[kPts];
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const float p2 = px[k] * px[k] + py[k] * py[k] + pz[k] * pz[k];
        const float S = p2 + v2max;
        // beyond the range the f16 operands cover comfortably every chunk becomes a candidate (-> exact whole-mesh scan)
        tol[k] = (S < 1.0e4f) ? 3.0517578e-5f * S : 3.0e38f;
        sb[k] = 3.f * tol[k] + p2;
        run[k] = 3.0e38f;
        sk[k] = 3.0e18f;                      // nothing is skipped before the first candidate
        cid[k] = 0ull;
        ncand[k] = 0;
#pragma unroll
        for (int q = 0; q < kCand; ++q) cval[k][q] = 3.4e38f;
    }
    const int n_chunks = Vpad / kChunk;
    // SORTED: the chunk whose sphere centre is nearest to the centroid of the wave's 256 points is scanned first
    int seed = -1;
    if constexpr (SORTED) {
        float sx = (px[0] + px[1]) + (px[2] + px[3]), sy = (py[0] + py[1]) + (py[2] + py[3]), sz = (pz[0] + pz[1]) + (pz[2] + pz[3]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            sx += __shfl_xor(sx, d, 64); sy += __shfl_xor(sy, d, 64); sz += __shfl_xor(sz, d, 64);
        }
        sx *= 1.f / 256.f; sy *= 1.f / 256.f; sz *= 1.f / 256.f;
        float bd = 3.4e38f;
        int bc = 0;
        for (int c0 = lane; c0 < n_chunks; c0 += 64) {
            const f4 cb = bounds[c0];
            const float dx = sx - cb.x, dy = sy - cb.y, dz = sz - cb.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < bd) { bd = d2; bc = c0; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float od = __shfl_xor(bd, d, 64);
            const int oc = __shfl_xor(bc, d, 64);
            if (od < bd || (od == bd && oc < bc)) { bd = od; bc = oc; }
        }
        seed = __builtin_amdgcn_readfirstlane(bc);
    }
#pragma clang loop unroll(disable)
    for (int it = 0; it < n_chunks; ++it) {
        // scan order: the seed chunk first, then the others in position order
        const int c = !SORTED ? it : it == 0 ? seed : (it <= seed ? it - 1 : it);
        if constexpr (SORTED) {
            // skip test (see the header): every point's search sphere misses the chunk's bounding sphere
            const f4 cb = bounds[c];                        // wave-uniform address: scalar loads
            bool need = false;
#pragma unroll
            for (int k = 0; k < kPts; ++k) {
                const float dx = px[k] - cb.x, dy = py[k] - cb.y, dz = pz[k] - cb.z;
                const float d2 = dx * dx + dy * dy + dz * dz;
                const float reach = cb.w + sk[k];
                need |= d2 <= reach * reach;
            }
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
        }
        float cmv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cmv[j] = 3.4e38f;
#pragma unroll
        for (int tile = 0; tile < kChunk / 32; ++tile) {
            const int v = c * kChunk + tile * 32 + m;             // this lane's A row
            const bool real = v < V;
            const float x = real ? vx[v] : 0.f, y = real ? vy[v] : 0.f, z = real ? vz[v] : 0.f;
            const float w = real ? fmaf(z, z, fmaf(y, y, x * x)) : 6.0e4f;      // padding rows: a = 60000, never minimal
            unsigned lo01, lo23;
            const unsigned hi01 = split2_f16(x, y, lo01), hi23 = split2_f16(z, w, lo23);
            const half8 afrag = __builtin_bit_cast(half8, hh ? u4{lo01, lo23, 0u, 0u} : u4{hi01, hi23, hi01, hi23});
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag[j], zero, 0, 0, 0);
                float mn = cmv[j];
#pragma unroll
                for (int r = 0; r < 16; r += 2) mn = fminf(fminf(mn, d[r]), d[r + 1]);
                cmv[j] = mn;
            }
        }
        float cm[kPts];
#pragma unroll
        for (int k = 0; k < kPts; ++k) {
            const float lo_set = fminf(cmv[k], __shfl_xor(cmv[k], 32, 64));
            const float hi_set = fminf(cmv[4 + k], __shfl_xor(cmv[4 + k], 32, 64));
            cm[k] = hh ? hi_set : lo_set;
        }
#pragma unroll
        for (int k = 0; k < kPts; ++k) {
            if (cm[k] <= run[k] + tol[k]) {           // rare: a (near-)record chunk -> remember it
#pragma unroll
                for (int q = 0; q < kCand - 1; ++q) cval[k][q] = cval[k][q + 1];
                cval[k][kCand - 1] = cm[k];
                cid[k] = (cid[k] << 8) | (unsigned long long)(c & 0xff);
                ++ncand[k];
                run[k] = fminf(run[k], cm[k]);
                if constexpr (SORTED)           // s^2 >= run + 3 tol + |p|^2, rounded up (tol = 3e38 -> inf: never skip)
                    sk[k] = __builtin_sqrtf(fmaxf(run[k] + sb[k], 0.f)) * 1.00001f;
            }
        }
    }
    // ---- refine: exact arithmetic over the surviving candidates, oldest (lowest chunk) first
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        // Candidates older than the remembered eight were dropped.  If a dropped one were still within tol of the
        // final minimum, every later candidate j (appended with a_j <= run_j + tol, run_j <= that one's value) would be
        // within 2 tol of it -- including the oldest remembered one.  So "oldest remembered > final + 2 tol" proves
        // nothing relevant was dropped; otherwise (eight near-ties in a row) scan the whole mesh exactly.
        const bool overflow = ncand[k] > kCand && cval[k][0] <= run[k] + 2.f * tol[k];
        if (overflow) {
            refine<SORTED>(vx4, vy4, vz4, vid4, 0, Vpad / 4, px[k], py[k], pz[k], best[k], bi[k]);
        } else {
#pragma unroll
            for (int q = 0; q < kCand; ++q) {
                if (cval[k][q] <= run[k] + tol[k]) {
                    const int c = (int)((cid[k] >> (8 * (kCand - 1 - q))) & 0xffull);
                    refine<SORTED>(vx4, vy4, vz4, vid4, c * (kChunk / 4), (c + 1) * (kChunk / 4), px[k], py[k], pz[k], best[k], bi[k]);
                }
            }
        }
    }

#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = point_of((4 * hh + k) * 32 + m);
        if (n >= N) continue;
        const int idx = bi[k];
        if constexpr (!FEATURES) {
            nn_index[(int64_t)b * N + n] = idx;
            continue;
        }
        const float4* __restrict__ M = reinterpret_cast<const float4*>(vertex_ik + ((int64_t)b * V + idx) * 16);
        const float4 r0 = M[0], r1 = M[1], r2 = M[2];
        const float x = px[k], y = py[k], z = pz[k];
        const float cx = (r0.x * x + r0.y * y + r0.z * z + r0.w) / 2.f;
        const float cy = ((r1.x * x + r1.y * y + r1.z * z + r1.w) + 0.2f) / 2.f;
        const float cz = (r2.x * x + r2.y * y + r2.z * z + r2.w) / 1.3f;
        const float* __restrict__ tv = tpose + ((int64_t)b * V + idx) * 3;
        float* o = geo + ((int64_t)b * N + n) * geo_stride;
        float* oc = legacy_mode ? o + kJoints : o;
        float* oj = legacy_mode ? o : o + 3;
        oc[0] = cx; oc[1] = cy; oc[2] = cz;
#pragma unroll
        for (int j = 0; j < kJoints; ++j) {
            const float ax = x - jl[j * 3 + 0], ay = y - jl[j * 3 + 1], az = z - jl[j * 3 + 2];
            oj[j] = sqrtf(ax * ax + ay * ay + az * az) / 2.4f;
        }
        o[27] = tv[0]; o[28] = tv[1]; o[29] = tv[2] / 0.2f;
        o[30] = sqrtf(best[k]) / 1.3f;
        if (nn_index) nn_index[(int64_t)b * N + n] = idx;
    }
}

// ---- h3d_mesh_sort: Morton order of a pose's vertices + bounding spheres of the chunks of 64 (one workgroup per pose)
constexpr int kSortThreads = 1024;

__device__ __forceinline__ unsigned spread3(unsigned q) {        // 10 bits -> every third bit
    q &= 0x3ffu;
    q = (q | (q << 16)) & 0x030000ffu;
    q = (q | (q << 8)) & 0x0300f00fu;
    q = (q | (q << 4)) & 0x030c30c3u;
    q = (q | (q << 2)) & 0x09249249u;
    return q;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ __launch_bounds__(kSortThreads) void mesh_sort_kernel(const float* __restrict__ vertices, f4* __restrict__ ws, int V, int Vpad,
                                                                 int NP2) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];        // NP2 keys
    __shared__ float box[6][kSortThreads / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* __restrict__ vb = vertices + (int64_t)b * V * 3;
    f4* __restrict__ out = ws + (int64_t)b * (Vpad + Vpad / kChunk);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = t; i < V; i += kSortThreads)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = vb[i * 3 + a];
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]);
        if (lane == 0) { box[a][wave] = lo[a]; box[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = box[a][0], h = box[3 + a][0];
        for (int w = 1; w < kSortThreads / 64; ++w) { l = fminf(l, box[a][w]); h = fmaxf(h, box[3 + a][w]); }
        lo[a] = l;
        inv[a] = h > l ? 1023.f / (h - l) : 0.f;
    }
    for (int i = t; i < NP2; i += kSortThreads) {
        unsigned long long key = ~0ull;                                  // padding sorts last
        if (i < V) {
            unsigned code = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float q = fminf(fmaxf((vb[i * 3 + a] - lo[a]) * inv[a], 0.f), 1023.f);
                code |= spread3((unsigned)q) << a;
            }
            key = ((unsigned long long)code << 32) | (unsigned)i;       // distinct keys: the order is a function of the mesh
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= NP2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < NP2; i += kSortThreads) {
                const int o = i ^ j;
                if (o > i) {
                    const unsigned long long a = keys[i], c = keys[o];
                    if ((a > c) == ((i & k) == 0)) { keys[i] = c; keys[o] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = t; i < Vpad; i += kSortThreads) {
        if (i < V) {
            const int idx = (int)(unsigned)(keys[i] & 0xffffffffull);
            out[i] = f4{vb[idx * 3 + 0], vb[idx * 3 + 1], vb[idx * 3 + 2], __int_as_float(idx)};
        } else {
            out[i] = f4{3.0e18f, 3.0e18f, 3.0e18f, __int_as_float(0x7fffffff)};       // never nearest, never wins a tie
        }
    }
    for (int c = wave; c < Vpad / kChunk; c += kSortThreads / 64) {     // every chunk holds at least one real vertex
        const int i = c * kChunk + lane;
        const bool real = i < V;
        const int idx = real ? (int)(unsigned)(keys[i] & 0xffffffffull) : 0;
        const float x = vb[idx * 3 + 0], y = vb[idx * 3 + 1], z = vb[idx * 3 + 2];
        const float cx = 0.5f * (wave_min(real ? x : 3.0e38f) + wave_max(real ? x : -3.0e38f));
        const float cy = 0.5f * (wave_min(real ? y : 3.0e38f) + wave_max(real ? y : -3.0e38f));
        const float cz = 0.5f * (wave_min(real ? z : 3.0e38f) + wave_max(real ? z : -3.0e38f));
        const float dx = x - cx, dy = y - cy, dz = z - cz;
        const float r2 = wave_max(real ? dx * dx + dy * dy + dz * dz : 0.f);
        if (lane == 0) out[Vpad + c] = f4{cx, cy, cz, __builtin_sqrtf(r2) * 1.00001f + 1.0e-7f};      // rounded up
    }
}

int next_pow2(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

static int geo_launch(bool features, bool sorted, const float* points, const float* joints, const float* vertices,
                      const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                      int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream, int Hr = 0, int Wr = 0, int S = 0) {
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "h3d_geo_features / h3d_nearest_vertex: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(V >= 1, "h3d_geo_features / h3d_nearest_vertex: V=%d", V);
    if (B == 0 || N == 0) return H3D_OK;
    const int Vpad = (V + kChunk - 1) / kChunk * kChunk;
    const size_t lds = sizeof(float) * ((sorted ? 4 : 3) * (size_t)Vpad + kJoints * 3 + 4);
    H3D_REQUIRE(lds <= 160 * 1024, "h3d_geo_features: mesh with V=%d vertices does not fit the 160 KB LDS", V);
    H3D_REQUIRE(!sorted || Vpad / kChunk <= 256, "h3d_geo_features: V=%d has more than 256 chunks", V);
    H3D_ALLOW_MAX_LDS((geo_features_kernel<true, false>));
    H3D_ALLOW_MAX_LDS((geo_features_kernel<false, false>));
    H3D_ALLOW_MAX_LDS((geo_features_kernel<true, true>));
    H3D_ALLOW_MAX_LDS((geo_features_kernel<false, true>));
    const int64_t per_block = (int64_t)kThreads * kPts;
    int64_t gx = (N + per_block - 1) / per_block;
    // compact wave tiles (8 x 8 rays x 4 samples) when the caller says how the points are laid out and the shape divides
    int tile_S = 0, tiles_x = 0, tiles_y = 0;
    if (S > 0 && Hr > 0 && Wr > 0 && (int64_t)Hr * Wr * S == N && Hr % 8 == 0 && Wr % 8 == 0 && S % 4 == 0) {
        tile_S = S; tiles_x = Wr / 8; tiles_y = Hr / 8;
        const int64_t waves = (int64_t)tiles_x * tiles_y * (S / 4);
        gx = (waves + kThreads / 64 - 1) / (kThreads / 64);
    }
    H3D_REQUIRE(gx < (int64_t(1) << 31), "h3d_geo_features: N too large");
    h3d::pre_launch();
    const dim3 grid((unsigned)gx, B), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define H3D_GEO_LAUNCH(F, S)                                                                                                    \
    hipLaunchKernelGGL((geo_features_kernel<F, S>), grid, block, lds, st, points, joints, vertices, tpose_vertices, vertex_ik, geo, \
                       nn_index, N, V, Vpad, geo_stride, legacy_mode, tile_S, Wr, tiles_x, tiles_y)
    if (features && sorted) H3D_GEO_LAUNCH(true, true);
    else if (features) H3D_GEO_LAUNCH(true, false);
    else if (sorted) H3D_GEO_LAUNCH(false, true);
    else H3D_GEO_LAUNCH(false, false);
#undef H3D_GEO_LAUNCH
    return h3d::launch_status(features ? "h3d_geo_features" : "h3d_nearest_vertex");
}

// Workspace of h3d_mesh_sort for B poses of V vertices, in bytes.
extern "C" int64_t h3d_mesh_sort_bytes(int B, int V) {
    if (B < 0 || V < 1) return -1;
    const int64_t Vpad = (V + kChunk - 1) / kChunk * kChunk;
    return (int64_t)B * (Vpad + Vpad / kChunk) * 16;
}

extern "C" int h3d_mesh_sort(const float* vertices, void* workspace, int B, int V, h3d_stream_t stream) {
    H3D_REQUIRE(vertices && workspace, "h3d_mesh_sort: null pointer");
    H3D_REQUIRE(B >= 0 && B <= 65535 && V >= 1, "h3d_mesh_sort: bad B=%d V=%d", B, V);
    H3D_REQUIRE(h3d::aligned16(workspace), "h3d_mesh_sort: the workspace must be 16-byte aligned");
    if (B == 0) return H3D_OK;
    const int Vpad = (V + kChunk - 1) / kChunk * kChunk, NP2 = next_pow2(V);
    H3D_REQUIRE((size_t)NP2 * 8 <= 128 * 1024, "h3d_mesh_sort: V=%d does not fit the LDS sort (max 16384)", V);
    H3D_ALLOW_MAX_LDS(mesh_sort_kernel);
    h3d::pre_launch();
    hipLaunchKernelGGL(mesh_sort_kernel, dim3(B), dim3(kSortThreads), (size_t)NP2 * 8, static_cast<hipStream_t>(stream), vertices,
                       static_cast<f4*>(workspace), V, Vpad, NP2);
    return h3d::launch_status("h3d_mesh_sort");
}

extern "C" int h3d_geo_features(const float* points, const float* joints, const float* vertices,
                                const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                                int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream) {
    H3D_REQUIRE(points && joints && vertices && tpose_vertices && vertex_ik && geo, "h3d_geo_features: null pointer");
    H3D_REQUIRE(geo_stride >= 31, "h3d_geo_features: geo_stride=%d must be >= 31", geo_stride);
    H3D_REQUIRE(h3d::aligned16(vertex_ik), "h3d_geo_features: vertex_ik must be 16-byte aligned");
    return geo_launch(true, false, points, joints, vertices, tpose_vertices, vertex_ik, geo, nn_index, B, N, V, geo_stride, legacy_mode, stream);
}

/* The K = 1 nearest-vertex search of h3d_geo_features alone (same filter + exact refine, same arg-min bit for bit):
 * nn_index [B, N] int32 is the only output.  The field kernels with GEOIN build the 31 features from it in their prologue
 * (h3d_render_fused_x2_geo / _x3_geo), so the [B, N, 31] feature tensor is neither written nor read back. */
extern "C" int h3d_nearest_vertex(const float* points, const float* vertices, int32_t* nn_index, int B, int64_t N, int V,
                                  h3d_stream_t stream) {
    H3D_REQUIRE(points && vertices && nn_index, "h3d_nearest_vertex: null pointer");
    return geo_launch(false, false, points, nullptr, vertices, nullptr, nullptr, nullptr, nn_index, B, N, V, 31, 0, stream);
}

/* The same two on a mesh prepared by h3d_mesh_sort (`sorted_mesh` = its workspace): chunks of 64 vertices whose bounding sphere
 * lies outside every search sphere of a wave's 256 points are skipped -- the same indices and features bit for bit (header of
 * this file), about a fifth of the chunks scanned. */
extern "C" int h3d_geo_features_sorted(const float* points, const float* joints, const void* sorted_mesh,
                                       const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                                       int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream) {
    H3D_REQUIRE(points && joints && sorted_mesh && tpose_vertices && vertex_ik && geo, "h3d_geo_features_sorted: null pointer");
    H3D_REQUIRE(geo_stride >= 31, "h3d_geo_features_sorted: geo_stride=%d must be >= 31", geo_stride);
    H3D_REQUIRE(h3d::aligned16(vertex_ik) && h3d::aligned16(sorted_mesh), "h3d_geo_features_sorted: vertex_ik / sorted_mesh must be 16-byte aligned");
    return geo_launch(true, true, points, joints, static_cast<const float*>(sorted_mesh), tpose_vertices, vertex_ik, geo, nn_index, B, N, V,
                      geo_stride, legacy_mode, stream);
}

/* h3d_nearest_vertex_sorted for points that are the samples of a render grid, points [B, Hr, Wr, S, 3] (rays row-major, the S
 * samples of a ray contiguous: what h3d_ray_setup writes): a wave then takes an 8 x 8 patch of rays x 4 consecutive samples
 * instead of 256 consecutive points (= a few whole rays), a compact box that lets the bounding-sphere test skip far more of the
 * mesh.  Same indices bit for bit.  Shapes that do not divide (Hr or Wr not a multiple of 8, S not a multiple of 4) fall back to
 * the linear assignment. */
extern "C" int h3d_nearest_vertex_sorted_rays(const float* points, const void* sorted_mesh, int32_t* nn_index, int B, int Hr, int Wr,
                                              int S, int V, h3d_stream_t stream) {
    H3D_REQUIRE(points && sorted_mesh && nn_index, "h3d_nearest_vertex_sorted_rays: null pointer");
    H3D_REQUIRE(h3d::aligned16(sorted_mesh), "h3d_nearest_vertex_sorted_rays: sorted_mesh must be 16-byte aligned");
    H3D_REQUIRE(Hr >= 1 && Wr >= 1 && S >= 1, "h3d_nearest_vertex_sorted_rays: bad grid %dx%dx%d", Hr, Wr, S);
    return geo_launch(false, true, points, nullptr, static_cast<const float*>(sorted_mesh), nullptr, nullptr, nullptr, nn_index, B,
                      (int64_t)Hr * Wr * S, V, 31, 0, stream, Hr, Wr, S);
}

extern "C" int h3d_nearest_vertex_sorted(const float* points, const void* sorted_mesh, int32_t* nn_index, int B, int64_t N, int V,
                                         h3d_stream_t stream) {
    H3D_REQUIRE(points && sorted_mesh && nn_index, "h3d_nearest_vertex_sorted: null pointer");
    H3D_REQUIRE(h3d::aligned16(sorted_mesh), "h3d_nearest_vertex_sorted: sorted_mesh must be 16-byte aligned");
    return geo_launch(false, true, points, nullptr, static_cast<const float*>(sorted_mesh), nullptr, nullptr, nullptr, nn_index, B, N, V, 31, 0,
                      stream);
}
