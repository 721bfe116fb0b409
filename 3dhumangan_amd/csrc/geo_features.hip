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
#include "common.hpp"

namespace {

constexpr int kThreads = 512;
constexpr int kPts = 4;
constexpr int kJoints = 24;
constexpr int kChunk = 64;                 // vertices per filter chunk
constexpr int kCand = 8;                   // remembered candidate chunks per point

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
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

// exact scan of vertices [v4_begin*4, v4_end*4) in ascending order, strict "<": first index among exact minima
__device__ __forceinline__ void refine(const f4* vx4, const f4* vy4, const f4* vz4, int v4_begin, int v4_end, float px,
                                       float py, float pz, float& best, int& bi) {
    for (int v4 = v4_begin; v4 < v4_end; ++v4) {
        const f4 X = vx4[v4], Y = vy4[v4], Z = vz4[v4];
        const float d0 = sqdist_exact(px, py, pz, X.x, Y.x, Z.x);
        const float d1 = sqdist_exact(px, py, pz, X.y, Y.y, Z.y);
        const float d2 = sqdist_exact(px, py, pz, X.z, Y.z, Z.z);
        const float d3 = sqdist_exact(px, py, pz, X.w, Y.w, Z.w);
        if (d0 < best) { best = d0; bi = v4 * 4 + 0; }
        if (d1 < best) { best = d1; bi = v4 * 4 + 1; }
        if (d2 < best) { best = d2; bi = v4 * 4 + 2; }
        if (d3 < best) { best = d3; bi = v4 * 4 + 3; }
    }
}

// FEATURES = false: the search only -- nn_index is the whole output (h3d_nearest_vertex; the features are then built in the field
// kernel's prologue, csrc/field_x3.hip GEOIN).
template <bool FEATURES>
__global__ __launch_bounds__(kThreads) void geo_features_kernel(
    const float* __restrict__ points, const float* __restrict__ joints, const float* __restrict__ vertices,
    const float* __restrict__ tpose, const float* __restrict__ vertex_ik, float* __restrict__ geo,
    int32_t* __restrict__ nn_index, int64_t N, int V, int Vpad, int geo_stride, int legacy_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* vx = smem;
    float* vy = smem + Vpad;
    float* vz = smem + 2 * Vpad;
    float* jl = smem + 3 * Vpad;   // 24*3 joints
    unsigned* v2max_bits = reinterpret_cast<unsigned*>(jl + kJoints * 3);
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const float* __restrict__ vb = vertices + (int64_t)b * V * 3;
    if (t == 0) *v2max_bits = 0u;
    __syncthreads();
    float v2 = 0.f;
    for (int i = t; i < Vpad; i += kThreads) {
        const bool ok = i < V;
        // padding vertices sit at +inf distance for the exact scan and can never win (the filter masks them itself)
        const float x = ok ? vb[i * 3 + 0] : 3.0e18f, y = ok ? vb[i * 3 + 1] : 3.0e18f, z = ok ? vb[i * 3 + 2] : 3.0e18f;
        vx[i] = x; vy[i] = y; vz[i] = z;
        if (ok) v2 = fmaxf(v2, x * x + y * y + z * z);
    }
    atomicMax(v2max_bits, __float_as_uint(v2));      // non-negative floats order like their bit patterns
    if (FEATURES && t < kJoints * 3) jl[t] = joints[(int64_t)b * kJoints * 3 + t];
    __syncthreads();
    const float v2max = __uint_as_float(*v2max_bits);

    // A wave owns 256 consecutive points as 8 sets of 32: lane (m, hh) owns points (4*hh + k) * 32 + m, k = 0..3.
    const int lane = t & 63, m = lane & 31, hh = lane >> 5;
    const int64_t wbase = ((int64_t)blockIdx.x * kThreads + (t & ~63)) * kPts;
    float px[kPts], py[kPts], pz[kPts], best[kPts];
    int bi[kPts];
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = wbase + (4 * hh + k) * 32 + m;
        const bool ok = n < N;
        const float* p = points + ((int64_t)b * N + (ok ? n : 0)) * 3;
        px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
        best[k] = 3.4e38f;
        bi[k] = 0;
    }
    const f4* vx4 = reinterpret_cast<const f4*>(vx);
    const f4* vy4 = reinterpret_cast<const f4*>(vy);
    const f4* vz4 = reinterpret_cast<const f4*>(vz);

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
    unsigned long long cid[kPts];
    int ncand[kPts];
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const float S = px[k] * px[k] + py[k] * py[k] + pz[k] * pz[k] + v2max;
        // beyond the range the f16 operands cover comfortably every chunk becomes a candidate (-> exact whole-mesh scan)
        tol[k] = (S < 1.0e4f) ? 3.0517578e-5f * S : 3.0e38f;
        run[k] = 3.0e38f;
        cid[k] = 0ull;
        ncand[k] = 0;
#pragma unroll
        for (int q = 0; q < kCand; ++q) cval[k][q] = 3.4e38f;
    }
    const int n_chunks = Vpad / kChunk;
    for (int c = 0; c < n_chunks; ++c) {
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
            refine(vx4, vy4, vz4, 0, Vpad / 4, px[k], py[k], pz[k], best[k], bi[k]);
        } else {
#pragma unroll
            for (int q = 0; q < kCand; ++q) {
                if (cval[k][q] <= run[k] + tol[k]) {
                    const int c = (int)((cid[k] >> (8 * (kCand - 1 - q))) & 0xffull);
                    refine(vx4, vy4, vz4, c * (kChunk / 4), (c + 1) * (kChunk / 4), px[k], py[k], pz[k], best[k], bi[k]);
                }
            }
        }
    }

#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = wbase + (4 * hh + k) * 32 + m;
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

}  // namespace

static int geo_launch(bool features, const float* points, const float* joints, const float* vertices,
                      const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                      int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream) {
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "h3d_geo_features / h3d_nearest_vertex: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(V >= 1, "h3d_geo_features / h3d_nearest_vertex: V=%d", V);
    if (B == 0 || N == 0) return H3D_OK;
    const int Vpad = (V + kChunk - 1) / kChunk * kChunk;
    const size_t lds = sizeof(float) * (3 * (size_t)Vpad + kJoints * 3 + 4);
    H3D_REQUIRE(lds <= 160 * 1024, "h3d_geo_features: mesh with V=%d vertices does not fit the 160 KB LDS", V);
    H3D_ALLOW_MAX_LDS(geo_features_kernel<true>);
    H3D_ALLOW_MAX_LDS(geo_features_kernel<false>);
    const int64_t per_block = (int64_t)kThreads * kPts;
    const int64_t gx = (N + per_block - 1) / per_block;
    H3D_REQUIRE(gx < (int64_t(1) << 31), "h3d_geo_features: N too large");
    h3d::pre_launch();
    if (features)
        hipLaunchKernelGGL(geo_features_kernel<true>, dim3((unsigned)gx, B), dim3(kThreads), lds, static_cast<hipStream_t>(stream),
                           points, joints, vertices, tpose_vertices, vertex_ik, geo, nn_index, N, V, Vpad, geo_stride,
                           legacy_mode);
    else
        hipLaunchKernelGGL(geo_features_kernel<false>, dim3((unsigned)gx, B), dim3(kThreads), lds, static_cast<hipStream_t>(stream),
                           points, joints, vertices, tpose_vertices, vertex_ik, geo, nn_index, N, V, Vpad, geo_stride,
                           legacy_mode);
    return h3d::launch_status(features ? "h3d_geo_features" : "h3d_nearest_vertex");
}

extern "C" int h3d_geo_features(const float* points, const float* joints, const float* vertices,
                                const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                                int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream) {
    H3D_REQUIRE(points && joints && vertices && tpose_vertices && vertex_ik && geo, "h3d_geo_features: null pointer");
    H3D_REQUIRE(geo_stride >= 31, "h3d_geo_features: geo_stride=%d must be >= 31", geo_stride);
    H3D_REQUIRE(h3d::aligned16(vertex_ik), "h3d_geo_features: vertex_ik must be 16-byte aligned");
    return geo_launch(true, points, joints, vertices, tpose_vertices, vertex_ik, geo, nn_index, B, N, V, geo_stride, legacy_mode, stream);
}

/* The K = 1 nearest-vertex search of h3d_geo_features alone (same filter + exact refine, same arg-min bit for bit):
 * nn_index [B, N] int32 is the only output.  The field kernels with GEOIN build the 31 features from it in their prologue
 * (h3d_render_fused_x2_geo / _x3_geo), so the [B, N, 31] feature tensor is neither written nor read back. */
extern "C" int h3d_nearest_vertex(const float* points, const float* vertices, int32_t* nn_index, int B, int64_t N, int V,
                                  h3d_stream_t stream) {
    H3D_REQUIRE(points && vertices && nn_index, "h3d_nearest_vertex: null pointer");
    return geo_launch(false, points, nullptr, vertices, nullptr, nullptr, nullptr, nn_index, B, N, V, 31, 0, stream);
}
