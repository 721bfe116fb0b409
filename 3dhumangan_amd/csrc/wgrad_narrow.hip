// Weight gradients with one NARROW side (training path, gfx950):  out[j][c] = sum_r narrow[r][j] * wide[r][c],  j < nn <= 4.
// These are the ToRGB layers (3 x C), the density / colour heads (1 x C, 3 x C) and the coordinate layer (C x 3, transposed by
// the caller): a 3 x 256 result from 0.5 M rows, for which the library GEMM needs 1.3 ms (0.4 TB/s).  Here the wide operand is
// streamed once with 16-byte loads (HBM-bound), the few narrow values of a row are broadcast loads, and every thread keeps
// nn x 4 running sums for its four channels.  Deterministic two-stage reduction: partial [nblk, nn, C], the caller sums nblk.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 256;                    // 2048 workgroups for 0.5 M rows: enough loads in flight to stream at HBM rate

// T = float (V = 4 or 1) or _Float16 (AMP tier, V = 8 or 1): both operands in T, products and sums in fp32
template <typename T, int V>
__global__ __launch_bounds__(kThreads) void wgrad_narrow_kernel(const T* __restrict__ wide, const T* __restrict__ narrow,
                                                                float* __restrict__ partial, int64_t M, int C, int ldw, int nn,
                                                                int ones, float* __restrict__ colsum) {
    // ones: one more output row (index nn, nn + ones <= 4) = the column sums of `wide` (a narrow column of ones): the bias
    // gradient when `wide` is dY.  colsum [nblk, 4]: the column sums of `narrow` over this block's rows: the bias gradient when
    // `narrow` is dY.  Either rides along the pass that is made anyway.
    const int no = nn + ones;
    __shared__ float red[4][kThreads][V];
    const int Q = C / V, QP = Q < kThreads ? Q : kThreads, G = kThreads / QP;
    const int t = threadIdx.x, g = t / QP;
    const int64_t r0 = (int64_t)blockIdx.x * kRows;
    const int64_t r1 = r0 + kRows < M ? r0 + kRows : M;
    float* out = partial + (int64_t)blockIdx.x * no * C;
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
                if constexpr (V == 1) {
                    v[0] = (float)wide[r * ldw + q];
                } else {                    // one 16-byte load: four floats or eight halves
                    typedef T vecT __attribute__((ext_vector_type(V)));
                    const vecT w = *reinterpret_cast<const vecT*>(wide + r * ldw + q * V);
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] = (float)w[k];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float nj = j < nn ? (float)narrow[r * nn + j] : (j < no ? 1.f : 0.f);
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
        // the row groups' sums, one output per thread and turn (round 6: all 256 threads; the first row group's QP threads alone
        // spent no * V * G dependent LDS reads each -- 256 at eight halves per thread -- while the others idled); same order of addition
        for (int o = t; o < no * QP * V; o += kThreads) {
            const int j = o / (QP * V), cc = o - j * (QP * V), qq = cc / V, k = cc - qq * V;
            if (q0 + qq < Q) {
                float s = 0.f;
                for (int gg = 0; gg < G; ++gg) s += red[j][gg * QP + qq][k];
                out[(int64_t)j * C + (q0 + qq) * V + k] = s;
            }
        }
        __syncthreads();
    }
    if (colsum != nullptr && t < 64) {           // first wave: lane l sums rows r0 + l, r0 + l + 64, ..; xor-shuffle fold
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t r = r0 + t; r < r1; r += 64)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nn) cs[j] += (float)narrow[r * nn + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) cs[j] += __shfl_xor(cs[j], d, 64);
            if (t == 0) colsum[(int64_t)blockIdx.x * 4 + j] = cs[j];
        }
    }
}

template <typename T, int V>
int launch_narrow(const T* wide, const T* narrow, float* partial, int64_t M, int C, int ldw, int nn, int ones, float* colsum,
                  hipStream_t st, const char* what) {
    const int64_t nblk = (M + kRows - 1) / kRows;
    h3d::pre_launch();
    hipLaunchKernelGGL((wgrad_narrow_kernel<T, V>), dim3((unsigned)nblk), dim3(kThreads), 0, st, wide, narrow, partial, M, C, ldw, nn,
                       ones, colsum);
    return h3d::launch_status(what);
}

int check_narrow(const void* wide, const void* narrow, const float* partial, int64_t M, int C, int ldw, int nn, int ones) {
    H3D_REQUIRE(wide && narrow && partial, "h3d_wgrad_narrow: null pointer");
    H3D_REQUIRE(M >= 1 && C >= 1 && ldw >= C, "h3d_wgrad_narrow: bad shape M=%lld C=%d ldw=%d", (long long)M, C, ldw);
    H3D_REQUIRE(nn >= 1 && nn <= 4, "h3d_wgrad_narrow: the narrow side must have 1..4 columns (got %d)", nn);
    H3D_REQUIRE((ones == 0 || ones == 1) && nn + ones <= 4, "h3d_wgrad_narrow: the extra row of column sums needs nn <= 3 (nn=%d)", nn);
    H3D_REQUIRE((M + kRows - 1) / kRows < (int64_t(1) << 31), "h3d_wgrad_narrow: too many rows");
    return 0;
}

}  // namespace

extern "C" int h3d_wgrad_narrow_rows(void) { return kRows; }

extern "C" int h3d_wgrad_narrow(const float* wide, const float* narrow, float* partial, int64_t M, int C, int ldw, int nn,
                                h3d_stream_t stream) {
    return h3d_wgrad_narrow_sums(wide, narrow, partial, nullptr, M, C, ldw, nn, 0, 0, stream);
}

extern "C" int h3d_wgrad_narrow_f16(const void* wide, const void* narrow, float* partial, int64_t M, int C, int ldw, int nn,
                                    h3d_stream_t stream) {
    return h3d_wgrad_narrow_sums(wide, narrow, partial, nullptr, M, C, ldw, nn, 0, 1, stream);
}

// The general form: operands fp32 (half = 0) or f16 (half = 1: the AMP tier -- the ToRGB / head / coordinate layers under float16
// autocast, where the library's f16 GEMM takes 4 ms for a 3 x 256 result from 0.5 M rows); `ones` = 1 appends the column sums of
// `wide` as output row nn (partial is then [nblk, nn + 1, C]); `colsum` (or null) receives [nblk, 4] column sums of `narrow`.
extern "C" int h3d_wgrad_narrow_sums(const void* wide, const void* narrow, float* partial, float* colsum, int64_t M, int C, int ldw,
                                     int nn, int ones, int half, h3d_stream_t stream) {
    if (int rc = check_narrow(wide, narrow, partial, M, C, ldw, nn, ones)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (half) {
        const _Float16* w = static_cast<const _Float16*>(wide);
        const _Float16* n = static_cast<const _Float16*>(narrow);
        if (C % 8 == 0 && ldw % 8 == 0 && h3d::aligned16(wide))
            return launch_narrow<_Float16, 8>(w, n, partial, M, C, ldw, nn, ones, colsum, st, "h3d_wgrad_narrow_f16");
        return launch_narrow<_Float16, 1>(w, n, partial, M, C, ldw, nn, ones, colsum, st, "h3d_wgrad_narrow_f16");
    }
    const float* w = static_cast<const float*>(wide);
    const float* n = static_cast<const float*>(narrow);
    if (C % 4 == 0 && ldw % 4 == 0 && h3d::aligned16(wide))
        return launch_narrow<float, 4>(w, n, partial, M, C, ldw, nn, ones, colsum, st, "h3d_wgrad_narrow");
    return launch_narrow<float, 1>(w, n, partial, M, C, ldw, nn, ones, colsum, st, "h3d_wgrad_narrow");
}
