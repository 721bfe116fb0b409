// Weight gradients with one NARROW side (training path, gfx950):  out[j][c] = sum_r narrow[r][j] * wide[r][c],  j < nn <= 4.
// These are the ToRGB layers (3 x C), the density / colour heads (1 x C, 3 x C) and the coordinate layer (C x 3, transposed by
// the caller): a 3 x 256 result from 0.5 M rows, for which the library GEMM needs 1.3 ms (0.4 TB/s).  Here the wide operand is
// streamed once with 16-byte loads (HBM-bound), the few narrow values of a row are broadcast loads, and every thread keeps
// nn x 4 running sums for its four channels.  Deterministic two-stage reduction: partial [nblk, nn, C], the caller sums nblk.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 256;                    // 2048 workgroups for 0.5 M rows: enough loads in flight to stream at HBM rate

template <int V>
__global__ __launch_bounds__(kThreads) void wgrad_narrow_kernel(const float* __restrict__ wide, const float* __restrict__ narrow,
                                                                float* __restrict__ partial, int64_t M, int C, int ldw, int nn) {
    __shared__ float red[4][kThreads][V];
    const int Q = C / V, QP = Q < kThreads ? Q : kThreads, G = kThreads / QP;
    const int t = threadIdx.x, g = t / QP;
    const int64_t r0 = (int64_t)blockIdx.x * kRows;
    const int64_t r1 = r0 + kRows < M ? r0 + kRows : M;
    float* out = partial + (int64_t)blockIdx.x * nn * C;
    for (int q0 = 0; q0 < Q; q0 += QP) {
        const int q = q0 + t - g * QP;
        float acc[4][V];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[j][k] = 0.f;
        if (g < G && q < Q) {
#pragma unroll 4
            for (int64_t r = r0 + g; r < r1; r += G) {
                float v[V];
                if constexpr (V == 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wide + r * ldw + q * 4);
                    v[0] = w4.x; v[1] = w4.y; v[2] = w4.z; v[3] = w4.w;
                } else {
                    v[0] = wide[r * ldw + q];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float nj = j < nn ? narrow[r * nn + j] : 0.f;
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[j][k] = fmaf(nj, v[k], acc[j][k]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < V; ++k) red[j][t][k] = acc[j][k];
        __syncthreads();
        if (g == 0 && q < Q) {
            for (int j = 0; j < nn; ++j)
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float s = 0.f;
                    for (int gg = 0; gg < G; ++gg) s += red[j][gg * QP + t][k];
                    out[(int64_t)j * C + q * V + k] = s;
                }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int h3d_wgrad_narrow_rows(void) { return kRows; }

extern "C" int h3d_wgrad_narrow(const float* wide, const float* narrow, float* partial, int64_t M, int C, int ldw, int nn,
                                h3d_stream_t stream) {
    H3D_REQUIRE(wide && narrow && partial, "h3d_wgrad_narrow: null pointer");
    H3D_REQUIRE(M >= 1 && C >= 1 && ldw >= C, "h3d_wgrad_narrow: bad shape M=%lld C=%d ldw=%d", (long long)M, C, ldw);
    H3D_REQUIRE(nn >= 1 && nn <= 4, "h3d_wgrad_narrow: the narrow side must have 1..4 columns (got %d)", nn);
    const int64_t nblk = (M + kRows - 1) / kRows;
    H3D_REQUIRE(nblk < (int64_t(1) << 31), "h3d_wgrad_narrow: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    h3d::pre_launch();
    if (C % 4 == 0 && ldw % 4 == 0 && h3d::aligned16(wide))
        hipLaunchKernelGGL(wgrad_narrow_kernel<4>, dim3((unsigned)nblk), dim3(kThreads), 0, st, wide, narrow, partial, M, C, ldw, nn);
    else
        hipLaunchKernelGGL(wgrad_narrow_kernel<1>, dim3((unsigned)nblk), dim3(kThreads), 0, st, wide, narrow, partial, M, C, ldw, nn);
    return h3d::launch_status("h3d_wgrad_narrow");
}
