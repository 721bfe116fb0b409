// Spectral normalisation of a convolution weight with one power iteration, for gfx950 (reference: torch.nn.utils.spectral_norm as
// the discriminator applies it, lib/discriminators/unet_discriminators.py:17 `norm_layer = torch.nn.utils.spectral_norm`; the
// arithmetic is SpectralNorm.compute_weight):
//     v' = normalize(W^T u),   u' = normalize(W v'),   sigma = u' . (W v'),   W_sn = W / sigma            W [R, K] row-major
// and its backward  dW = (G - c u' v'^T) / sigma,  c = sum(G * W_sn)  (sigma = u'^T W v' with u', v' constants).
// torch spends ~13 launches per layer and forward on this (two mv, two norms, clamps, divisions, clones, mv, dot, a division of
// the whole weight) and ~8 on its backward; the discriminator has 25 such layers and runs three times per iteration -- ~16 ms of
// 220 in tiny kernels.  Here: four launches forward, two backward, every reduction two-stage in a fixed order (deterministic).
//   h3d_spectral_norm      sn_wtu (partial W^T u per 16 rows) -> sn_treduce (t, partial |t|^2) -> sn_wv (v' out, s = W v', partial |s|^2) -> sn_scale (u', sigma, W_sn)
//   h3d_spectral_norm_bwd  sn_dot (partial sum(G * W_sn)) -> sn_bwd (dW)
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxParts = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// block-wide sum of one value per thread (fixed order): returns the total to every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ float sum_parts(const float* __restrict__ parts, int n, float* red) {
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += kThreads) v += parts[i];
    return block_sum(v, red);
}

// tp[p][j] = sum over the p-th group of kRowsPer rows of W[i, j] u[i]: a thread per column, a block per (256 columns, row group):
// R / 16 x K / 256 workgroups (a single workgroup per 256 columns walking all R rows is latency-bound: 100 us for 512 x 4608)
constexpr int kRowsPer = 16;
__global__ __launch_bounds__(kThreads) void sn_wtu(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ tp,
                                                   int R, int K) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    const int i0 = blockIdx.y * kRowsPer;
    if (j >= K) return;
    float w[kRowsPer];
#pragma unroll
    for (int r = 0; r < kRowsPer; ++r) w[r] = i0 + r < R ? W[(int64_t)(i0 + r) * K + j] : 0.f;      // all loads in flight
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < kRowsPer; ++r) acc = fmaf(w[r], i0 + r < R ? u[i0 + r] : 0.f, acc);
    tp[(int64_t)blockIdx.y * K + j] = acc;
}

// t[j] = sum_p tp[p][j] (fixed order); parts[block] = sum of t^2 over the block's columns
__global__ __launch_bounds__(kThreads) void sn_treduce(const float* __restrict__ tp, int P, float* __restrict__ t, float* __restrict__ parts,
                                                       int K) {
    __shared__ float red[kThreads / 64];
    const int j = blockIdx.x * kThreads + threadIdx.x;
    float acc = 0.f;
    if (j < K)
        for (int p = 0; p < P; ++p) acc += tp[(int64_t)p * K + j];
    if (j < K) t[j] = acc;
    const float q = block_sum(j < K ? acc * acc : 0.f, red);
    if (threadIdx.x == 0) parts[blockIdx.x] = q;
}

// v' = t / max(|t|, eps) (written by block 0 into both destinations); s[i] = W[i, :] . v': a wave per row;
// parts2[block] = sum of s^2 over the block's rows
__global__ __launch_bounds__(kThreads) void sn_wv(const float* __restrict__ W, const float* __restrict__ t, const float* __restrict__ parts,
                                                  int n_parts, float eps, float* __restrict__ v_out, float* __restrict__ v_buf,
                                                  float* __restrict__ s, float* __restrict__ parts2, int R, int K) {
    __shared__ float red[kThreads / 64];
    const float nt = __builtin_sqrtf(sum_parts(parts, n_parts, red));
    const float inv = 1.f / fmaxf(nt, eps);
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < K; j += kThreads) {
            const float v = t[j] * inv;
            v_out[j] = v;
            v_buf[j] = v;
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * (kThreads / 64) + wave;
    float acc = 0.f;
    if (i < R)
        for (int j = lane; j < K; j += 64) acc = fmaf(W[(int64_t)i * K + j], t[j] * inv, acc);
    acc = wave_sum(acc);
    if (i < R && lane == 0) s[i] = acc;
    __syncthreads();
    if (lane == 0) red[wave] = i < R ? acc * acc : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float p = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) p += red[w];
        parts2[blockIdx.x] = p;
    }
}

// u' = s / max(|s|, eps), sigma = u' . s; W_sn = W / sigma.  Block 0 also writes u' (both destinations) and sigma.
__global__ __launch_bounds__(kThreads) void sn_scale(const float* __restrict__ W, const float* __restrict__ s, const float* __restrict__ parts2,
                                                     int n_parts2, float eps, float* __restrict__ u_out, float* __restrict__ u_buf,
                                                     float* __restrict__ sigma_out, float* __restrict__ Wsn, int R, int64_t n) {
    __shared__ float red[kThreads / 64];
    const float ss = sum_parts(parts2, n_parts2, red);
    const float inv = 1.f / fmaxf(__builtin_sqrtf(ss), eps);
    const float sigma = ss * inv;                                  // sum_i (s_i inv) s_i
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < R; i += kThreads) {
            const float u = s[i] * inv;
            u_out[i] = u;
            u_buf[i] = u;
        }
        if (threadIdx.x == 0) *sigma_out = sigma;
    }
    for (int64_t e = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4; e < n; e += (int64_t)gridDim.x * kThreads * 4) {
        if (e + 3 < n) {
            const float4 w = *reinterpret_cast<const float4*>(W + e);
            *reinterpret_cast<float4*>(Wsn + e) = make_float4(w.x / sigma, w.y / sigma, w.z / sigma, w.w / sigma);
        } else {
            for (int64_t q = e; q < n; ++q) Wsn[q] = W[q] / sigma;
        }
    }
}

__global__ __launch_bounds__(kThreads) void sn_dot(const float* __restrict__ G, const float* __restrict__ Wsn, float* __restrict__ parts,
                                                   int64_t n) {
    __shared__ float red[kThreads / 64];
    float acc = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads) acc = fmaf(G[e], Wsn[e], acc);
    const float p = block_sum(acc, red);
    if (threadIdx.x == 0) parts[blockIdx.x] = p;
}

__global__ __launch_bounds__(kThreads) void sn_bwd(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                                                   const float* __restrict__ sigma, const float* __restrict__ parts, int n_parts,
                                                   float* __restrict__ dW, int K, int64_t n) {
    __shared__ float red[kThreads / 64];
    const float c = sum_parts(parts, n_parts, red);
    const float sg = *sigma;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kThreads) {
        const int i = (int)(e / K), j = (int)(e - (int64_t)i * K);
        dW[e] = (G[e] - c * u[i] * v[j]) / sg;
    }
}

int grid_for(int64_t n, int per_thread) {
    int64_t b = (n + (int64_t)kThreads * per_thread - 1) / ((int64_t)kThreads * per_thread);
    return (int)(b < 1 ? 1 : b > kMaxParts ? kMaxParts : b);
}

}  // namespace

// Scratch floats h3d_spectral_norm needs for an [R, K] weight: t [K], s [R], two partial arrays.
extern "C" int64_t h3d_spectral_norm_scratch(int R, int K) {
    if (R < 1 || K < 1) return -1;
    const int64_t P = (R + kRowsPer - 1) / kRowsPer;
    return (int64_t)K * (P + 1) + R + 2 * kMaxParts;
}

extern "C" int h3d_spectral_norm(const float* W, const float* u, float* u_out, float* u_buf, float* v_out, float* v_buf, float* sigma,
                                 float* W_sn, float* scratch, int R, int K, float eps, h3d_stream_t stream) {
    H3D_REQUIRE(W && u && u_out && u_buf && v_out && v_buf && sigma && W_sn && scratch, "h3d_spectral_norm: null pointer");
    H3D_REQUIRE(R >= 1 && K >= 1, "h3d_spectral_norm: bad shape R=%d K=%d", R, K);
    H3D_REQUIRE(h3d::aligned16(W) && h3d::aligned16(W_sn), "h3d_spectral_norm: W / W_sn must be 16-byte aligned");
    const int nb1 = (K + kThreads - 1) / kThreads, nb2 = (R + kThreads / 64 - 1) / (kThreads / 64);
    H3D_REQUIRE(nb1 <= kMaxParts && nb2 <= kMaxParts, "h3d_spectral_norm: weight too large (R=%d, K=%d)", R, K);
    const int P = (R + kRowsPer - 1) / kRowsPer;
    float *t = scratch, *s = scratch + K, *p1 = s + R, *p2 = p1 + kMaxParts, *tp = p2 + kMaxParts;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)R * K;
    h3d::pre_launch();
    hipLaunchKernelGGL(sn_wtu, dim3(nb1, P), dim3(kThreads), 0, st, W, u, tp, R, K);
    hipLaunchKernelGGL(sn_treduce, dim3(nb1), dim3(kThreads), 0, st, tp, P, t, p1, K);
    hipLaunchKernelGGL(sn_wv, dim3(nb2), dim3(kThreads), 0, st, W, t, p1, nb1, eps, v_out, v_buf, s, p2, R, K);
    hipLaunchKernelGGL(sn_scale, dim3(grid_for(n, 4)), dim3(kThreads), 0, st, W, s, p2, nb2, eps, u_out, u_buf, sigma, W_sn, R, n);
    return h3d::launch_status("h3d_spectral_norm");
}

// dW = (G - sum(G * W_sn) u v^T) / sigma;  scratch: kMaxParts floats (h3d_spectral_norm_scratch covers it).
extern "C" int h3d_spectral_norm_bwd(const float* G, const float* W_sn, const float* u, const float* v, const float* sigma, float* dW,
                                     float* scratch, int R, int K, h3d_stream_t stream) {
    H3D_REQUIRE(G && W_sn && u && v && sigma && dW && scratch, "h3d_spectral_norm_bwd: null pointer");
    H3D_REQUIRE(R >= 1 && K >= 1, "h3d_spectral_norm_bwd: bad shape R=%d K=%d", R, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)R * K;
    const int nb = grid_for(n, 8);
    h3d::pre_launch();
    hipLaunchKernelGGL(sn_dot, dim3(nb), dim3(kThreads), 0, st, G, W_sn, scratch, n);
    hipLaunchKernelGGL(sn_bwd, dim3(grid_for(n, 4)), dim3(kThreads), 0, st, G, u, v, sigma, scratch, nb, dW, K, n);
    return h3d::launch_status("h3d_spectral_norm_bwd");
}
