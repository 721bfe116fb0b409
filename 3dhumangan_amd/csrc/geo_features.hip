// A4 SMPL geometry features for gfx950.
// Reference semantics: lib/components/smpl.py:210-249 (get_geo_features); K=1 nearest vertex per
// pytorch3d.ops.knn_points contract (squared L2, first index wins exact ties).
//
// Workgroup = 512 threads, 2 points per thread.  The pose's mesh (V x 3 fp32 = 83 KB for SMPL) is staged once
// per workgroup into LDS as three SoA planes; every lane then sweeps all V vertices, 4 per step, reading the
// planes with wave-uniform (broadcast) ds_read_b128.  Squared distance is evaluated exactly as the oracle does
// -- (dx*dx + dy*dy) + dz*dz with no fused multiply-add -- so the arg-min is bit-for-bit reproducible.
// The per-point tail gathers the blended inverse bone transform of the winner (64 B), canonicalises the point,
// gathers the T-pose vertex and evaluates the 24 joint distances.
#include "common.hpp"

namespace {

constexpr int kThreads = 512;
constexpr int kPts = 2;
constexpr int kJoints = 24;

__device__ __forceinline__ float sqdist_exact(float px, float py, float pz, float vx, float vy, float vz) {
    const float dx = __fsub_rn(px, vx), dy = __fsub_rn(py, vy), dz = __fsub_rn(pz, vz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__global__ __launch_bounds__(kThreads) void geo_features_kernel(
    const float* __restrict__ points, const float* __restrict__ joints, const float* __restrict__ vertices,
    const float* __restrict__ tpose, const float* __restrict__ vertex_ik, float* __restrict__ geo,
    int32_t* __restrict__ nn_index, int64_t N, int V, int Vpad, int geo_stride, int legacy_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* vx = smem;
    float* vy = smem + Vpad;
    float* vz = smem + 2 * Vpad;
    float* jl = smem + 3 * Vpad;   // 24*3 joints
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const float* __restrict__ vb = vertices + (int64_t)b * V * 3;
    for (int i = t; i < Vpad; i += kThreads) {
        const bool ok = i < V;
        // padding vertices sit at +inf distance and can never win
        vx[i] = ok ? vb[i * 3 + 0] : 3.0e18f;
        vy[i] = ok ? vb[i * 3 + 1] : 3.0e18f;
        vz[i] = ok ? vb[i * 3 + 2] : 3.0e18f;
    }
    if (t < kJoints * 3) jl[t] = joints[(int64_t)b * kJoints * 3 + t];
    __syncthreads();

    const int64_t base = ((int64_t)blockIdx.x * kThreads + t) * kPts;
    float px[kPts], py[kPts], pz[kPts], best[kPts];
    int bi[kPts];
#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = base + k;
        const bool ok = n < N;
        const float* p = points + ((int64_t)b * N + (ok ? n : 0)) * 3;
        px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
        best[k] = 3.4e38f;
        bi[k] = 0;
    }
    const float4* vx4 = reinterpret_cast<const float4*>(vx);
    const float4* vy4 = reinterpret_cast<const float4*>(vy);
    const float4* vz4 = reinterpret_cast<const float4*>(vz);
    for (int v4 = 0; v4 < Vpad / 4; ++v4) {
        const float4 X = vx4[v4], Y = vy4[v4], Z = vz4[v4];
#pragma unroll
        for (int k = 0; k < kPts; ++k) {
            const float d0 = sqdist_exact(px[k], py[k], pz[k], X.x, Y.x, Z.x);
            const float d1 = sqdist_exact(px[k], py[k], pz[k], X.y, Y.y, Z.y);
            const float d2 = sqdist_exact(px[k], py[k], pz[k], X.z, Y.z, Z.z);
            const float d3 = sqdist_exact(px[k], py[k], pz[k], X.w, Y.w, Z.w);
            if (d0 < best[k]) { best[k] = d0; bi[k] = v4 * 4 + 0; }
            if (d1 < best[k]) { best[k] = d1; bi[k] = v4 * 4 + 1; }
            if (d2 < best[k]) { best[k] = d2; bi[k] = v4 * 4 + 2; }
            if (d3 < best[k]) { best[k] = d3; bi[k] = v4 * 4 + 3; }
        }
    }

#pragma unroll
    for (int k = 0; k < kPts; ++k) {
        const int64_t n = base + k;
        if (n >= N) continue;
        const int idx = bi[k];
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

extern "C" int h3d_geo_features(const float* points, const float* joints, const float* vertices,
                                const float* tpose_vertices, const float* vertex_ik, float* geo, int32_t* nn_index,
                                int B, int64_t N, int V, int geo_stride, int legacy_mode, h3d_stream_t stream) {
    H3D_REQUIRE(points && joints && vertices && tpose_vertices && vertex_ik && geo, "h3d_geo_features: null pointer");
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "h3d_geo_features: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(V >= 1, "h3d_geo_features: V=%d", V);
    H3D_REQUIRE(geo_stride >= 31, "h3d_geo_features: geo_stride=%d must be >= 31", geo_stride);
    H3D_REQUIRE(h3d::aligned16(vertex_ik), "h3d_geo_features: vertex_ik must be 16-byte aligned");
    if (B == 0 || N == 0) return H3D_OK;
    const int Vpad = (V + 3) & ~3;
    const size_t lds = sizeof(float) * (3 * (size_t)Vpad + kJoints * 3);
    H3D_REQUIRE(lds <= 160 * 1024, "h3d_geo_features: mesh with V=%d vertices does not fit the 160 KB LDS", V);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_features_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    const int64_t per_block = (int64_t)kThreads * kPts;
    const int64_t gx = (N + per_block - 1) / per_block;
    H3D_REQUIRE(gx < (int64_t(1) << 31), "h3d_geo_features: N too large");
    h3d::pre_launch();
    hipLaunchKernelGGL(geo_features_kernel, dim3((unsigned)gx, B), dim3(kThreads), lds, static_cast<hipStream_t>(stream),
                       points, joints, vertices, tpose_vertices, vertex_ik, geo, nn_index, N, V, Vpad, geo_stride,
                       legacy_mode);
    return h3d::launch_status("h3d_geo_features");
}
