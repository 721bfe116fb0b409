// Training-side SPADE kernels for gfx950 (backward of A9): BatchNorm + SPADE modulation + LeakyReLU
//     y = lrelu( ((x - mean) * rstd * g + b) * (1 + gamma) + beta )
// (lib/components/map3d_layers.py:176-190 followed by the block's activation, :228-233) as ONE pass forward and TWO passes
// backward over channels-last activations x [B, P, C], instead of the ~12 elementwise / reduction passes torch's autograd
// spends on the same chain.  gamma / beta are per pixel [B, P, C] or per sample [B, C] (constant-style SPADE).  All HBM-bound.
//
//   h3d_channel_moments      per-workgroup sums of x and x^2 per channel            (batch statistics; caller reduces)
//   h3d_spade_fwd            the forward pass
//   h3d_spade_bwd_reduce     per-workgroup sums of dh and dh * n per channel, dh = dL/d(normalised-affine output)
//   h3d_spade_bwd_apply      dx (with the batch-statistics terms), dgamma, dbeta (tensors, or per-workgroup sums per sample)
//
// Thread layout (all four): a workgroup owns `rows` consecutive pixels of one batch item; thread t owns V consecutive channels
// (q = t % QP) of the rows g, g + G, ... (g = t / QP), so per-channel constants stay in registers and consecutive lanes touch
// consecutive 16 bytes.  Partial sums leave through LDS in a fixed order: the two-stage reductions are deterministic.
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 512;

template <int V> __device__ __forceinline__ void ld(const float* p, float (&v)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = p[0];
    }
}
template <int V> __device__ __forceinline__ void st(float* p, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else p[0] = v[0];
}
// f16 storage (the AMP tier, round 4): the big tensors -- x, y, dy, dx and the per-pixel gamma / beta and their gradients -- travel as
// _Float16, 8 bytes per 4 channels; every value is widened on load and the arithmetic stays fp32 in registers, rounded once at the store
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V> __device__ __forceinline__ void ld(const _Float16* p, float (&v)[V]) {
    if constexpr (V == 4) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        const h16x2 a = __builtin_bit_cast(h16x2, u.x), b = __builtin_bit_cast(h16x2, u.y);
        v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b.x; v[3] = (float)b.y;
    } else {
        v[0] = (float)p[0];
    }
}
template <int V> __device__ __forceinline__ void st(_Float16* p, const float (&v)[V]) {
    if constexpr (V == 4) {
        const h16x2 a = __builtin_convertvector(f32x2{v[0], v[1]}, h16x2), b = __builtin_convertvector(f32x2{v[2], v[3]}, h16x2);
        *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    } else {
        p[0] = (_Float16)v[0];
    }
}
// gamma / beta: per-pixel tensors of the activations' type, or per-sample fp32 vectors
template <bool PIX, typename T> struct Mod { typedef typename std::conditional<PIX, T, float>::type type; };

struct Tile {
    int Q, QP, G, g, b;
    int64_t r0, r1;
    template <int V> __device__ __forceinline__ void init(int64_t P, int C) {
        Q = C / V;
        QP = Q < kThreads ? Q : kThreads;
        G = kThreads / QP;
        g = threadIdx.x / QP;
        b = blockIdx.y;
        r0 = (int64_t)blockIdx.x * kRows;
        r1 = r0 + kRows < P ? r0 + kRows : P;
    }
};

// two per-channel sums of this workgroup -> out[0][c], out[1][c]   (every thread must call it)
template <int V>
__device__ __forceinline__ void block_sums(float (*red)[kThreads][V], const Tile& T, int q, const float (&a)[V],
                                           const float (&c)[V], float* out, int C) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < V; ++k) { red[0][t][k] = a[k]; red[1][t][k] = c[k]; }
    __syncthreads();
    if (T.g == 0 && q < T.Q) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float s0 = 0.f, s1 = 0.f;
            for (int gg = 0; gg < T.G; ++gg) { s0 += red[0][gg * T.QP + t][k]; s1 += red[1][gg * T.QP + t][k]; }
            out[q * V + k] = s0;
            out[C + q * V + k] = s1;
        }
    }
    __syncthreads();
}

template <int V, typename TA>
__global__ __launch_bounds__(kThreads) void channel_moments(const TA* __restrict__ x, float* __restrict__ partial,
                                                            int64_t P, int C) {
    __shared__ float red[2][kThreads][V];
    Tile T;
    T.init<V>(P, C);
    float* out = partial + ((int64_t)T.b * gridDim.x + blockIdx.x) * 2 * C;
    for (int q0 = 0; q0 < T.Q; q0 += T.QP) {
        const int q = q0 + threadIdx.x - T.g * T.QP;
        float s[V], ss[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = ss[k] = 0.f;
        if (T.g < T.G && q < T.Q)
            for (int64_t r = T.r0 + T.g; r < T.r1; r += T.G) {
                float v[V];
                ld<V>(x + ((int64_t)T.b * P + r) * C + q * V, v);
#pragma unroll
                for (int k = 0; k < V; ++k) { s[k] += v[k]; ss[k] = fmaf(v[k], v[k], ss[k]); }
            }
        block_sums<V>(red, T, q, s, ss, out, C);
    }
}

// scale = rstd * g, shift = b - mean * scale
template <int V, bool PIX, typename TA>
__global__ __launch_bounds__(kThreads) void spade_fwd(const TA* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const typename Mod<PIX, TA>::type* __restrict__ gamma,
                                                      const typename Mod<PIX, TA>::type* __restrict__ beta, TA* __restrict__ y, int64_t P, int C,
                                                      float slope) {
    Tile T;
    T.init<V>(P, C);
    if (T.g >= T.G) return;
    for (int q = threadIdx.x - T.g * T.QP; q < T.Q; q += T.QP) {
        float sc[V], sh[V], ga[V], be[V];
        ld<V>(scale + q * V, sc);
        ld<V>(shift + q * V, sh);
        if (!PIX) {
            ld<V>(gamma + (int64_t)T.b * C + q * V, ga);
            ld<V>(beta + (int64_t)T.b * C + q * V, be);
        }
        for (int64_t r = T.r0 + T.g; r < T.r1; r += T.G) {
            const int64_t off = ((int64_t)T.b * P + r) * C + q * V;
            float v[V];
            ld<V>(x + off, v);
            if (PIX) { ld<V>(gamma + off, ga); ld<V>(beta + off, be); }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float u = fmaf(fmaf(v[k], sc[k], sh[k]), 1.f + ga[k], be[k]);
                v[k] = u > 0.f ? u : u * slope;
            }
            st<V>(y + off, v);
        }
    }
}

// partial[b][blk][0][c] = sum dh, [1][c] = sum dh * n   with n = (x - mean) * rstd, h = n * g + b, u = h (1 + gamma) + beta,
// du = dy * lrelu'(u), dh = du * (1 + gamma)
template <int V, bool PIX, typename TA>
__global__ __launch_bounds__(kThreads) void spade_bwd_reduce(const TA* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gw,
                                                             const float* __restrict__ gb, const typename Mod<PIX, TA>::type* __restrict__ gamma,
                                                             const typename Mod<PIX, TA>::type* __restrict__ beta, const TA* __restrict__ dy,
                                                             float* __restrict__ partial, int64_t P, int C, float slope) {
    __shared__ float red[2][kThreads][V];
    Tile T;
    T.init<V>(P, C);
    float* out = partial + ((int64_t)T.b * gridDim.x + blockIdx.x) * 2 * C;
    for (int q0 = 0; q0 < T.Q; q0 += T.QP) {
        const int q = q0 + threadIdx.x - T.g * T.QP;
        float s1[V], s2[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
        if (T.g < T.G && q < T.Q) {
            float mu[V], rs[V], w[V], bb[V], ga[V], be[V];
            ld<V>(mean + q * V, mu);
            ld<V>(rstd + q * V, rs);
            ld<V>(gw + q * V, w);
            ld<V>(gb + q * V, bb);
            if (!PIX) {
                ld<V>(gamma + (int64_t)T.b * C + q * V, ga);
                ld<V>(beta + (int64_t)T.b * C + q * V, be);
            }
            for (int64_t r = T.r0 + T.g; r < T.r1; r += T.G) {
                const int64_t off = ((int64_t)T.b * P + r) * C + q * V;
                float v[V], d[V];
                ld<V>(x + off, v);
                ld<V>(dy + off, d);
                if (PIX) { ld<V>(gamma + off, ga); ld<V>(beta + off, be); }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float n = (v[k] - mu[k]) * rs[k];
                    const float u = fmaf(fmaf(n, w[k], bb[k]), 1.f + ga[k], be[k]);
                    const float dh = (u > 0.f ? d[k] : d[k] * slope) * (1.f + ga[k]);
                    s1[k] += dh;
                    s2[k] = fmaf(dh, n, s2[k]);
                }
            }
        }
        block_sums<V>(red, T, q, s1, s2, out, C);
    }
}

// dx = rstd * g * (dh - c1 - n * c2)   (c1 = sum dh / M, c2 = sum dh n / M for batch statistics; zeros for running statistics)
// PIX: dgamma = du * h, dbeta = du as tensors.  !PIX: their per-workgroup sums -> partial[b][blk][0|1][c].
template <int V, bool PIX, typename TA>
__global__ __launch_bounds__(kThreads) void spade_bwd_apply(const TA* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gw,
                                                            const float* __restrict__ gb, const typename Mod<PIX, TA>::type* __restrict__ gamma,
                                                            const typename Mod<PIX, TA>::type* __restrict__ beta, const TA* __restrict__ dy,
                                                            const float* __restrict__ c1, const float* __restrict__ c2,
                                                            TA* __restrict__ dx, TA* __restrict__ dgamma,
                                                            TA* __restrict__ dbeta, float* __restrict__ partial, int64_t P,
                                                            int C, float slope, const TA* __restrict__ add1, const TA* __restrict__ add2) {
    __shared__ float red[2][kThreads][V];
    Tile T;
    T.init<V>(P, C);
    float* out = PIX ? nullptr : partial + ((int64_t)T.b * gridDim.x + blockIdx.x) * 2 * C;
    for (int q0 = 0; q0 < T.Q; q0 += T.QP) {
        const int q = q0 + threadIdx.x - T.g * T.QP;
        float sg[V], sb[V];
#pragma unroll
        for (int k = 0; k < V; ++k) sg[k] = sb[k] = 0.f;
        if (T.g < T.G && q < T.Q) {
            float mu[V], rs[V], w[V], bb[V], ga[V], be[V], k1[V], k2[V];
            ld<V>(mean + q * V, mu);
            ld<V>(rstd + q * V, rs);
            ld<V>(gw + q * V, w);
            ld<V>(gb + q * V, bb);
            ld<V>(c1 + q * V, k1);
            ld<V>(c2 + q * V, k2);
            if (!PIX) {
                ld<V>(gamma + (int64_t)T.b * C + q * V, ga);
                ld<V>(beta + (int64_t)T.b * C + q * V, be);
            }
            for (int64_t r = T.r0 + T.g; r < T.r1; r += T.G) {
                const int64_t off = ((int64_t)T.b * P + r) * C + q * V;
                float v[V], d[V], o_g[V], o_b[V];
                ld<V>(x + off, v);
                ld<V>(dy + off, d);
                if (PIX) { ld<V>(gamma + off, ga); ld<V>(beta + off, be); }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float n = (v[k] - mu[k]) * rs[k];
                    const float h = fmaf(n, w[k], bb[k]);
                    const float u = fmaf(h, 1.f + ga[k], be[k]);
                    const float du = u > 0.f ? d[k] : d[k] * slope;
                    const float dh = du * (1.f + ga[k]);
                    v[k] = rs[k] * w[k] * (dh - k1[k] - n * k2[k]);
                    o_g[k] = du * h;
                    o_b[k] = du;
                    sg[k] += o_g[k];
                    sb[k] += du;
                }
                // gradients that reach x along other edges of the graph (a residual connection, a ToRGB head) join here instead of
                // in accumulation passes of their own (h3d_spade_bwd_apply_acc)
                if (add1) {
                    float a[V];
                    ld<V>(add1 + off, a);
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] += a[k];
                }
                if (add2) {
                    float a[V];
                    ld<V>(add2 + off, a);
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] += a[k];
                }
                st<V>(dx + off, v);
                if (PIX) { st<V>(dgamma + off, o_g); st<V>(dbeta + off, o_b); }
            }
        }
        if (!PIX) block_sums<V>(red, T, q, sg, sb, out, C);
    }
}

bool vec_ok(int C, std::initializer_list<const void*> ptrs) {
    if (C % 4) return false;
    for (const void* p : ptrs)
        if (p && !h3d::aligned16(p)) return false;
    return true;
}

int check_shape(const char* what, int B, int64_t P, int C) {
    if (B < 0 || P < 0 || C < 1 || B > 65535) {
        h3d::set_error("%s: bad shape B=%d P=%lld C=%d", what, B, (long long)P, C);
        return H3D_EINVAL;
    }
    return H3D_OK;
}

}  // namespace

extern "C" int h3d_spade_rows(void) { return kRows; }

template <typename T>
static int channel_moments_any(const T* x, float* partial, int B, int64_t P, int C, h3d_stream_t stream) {
    if (int rc = check_shape("h3d_channel_moments", B, P, C)) return rc;
    if (B == 0 || P == 0) return H3D_OK;
    H3D_REQUIRE(x && partial, "h3d_channel_moments: null pointer");
    const dim3 grid((unsigned)((P + kRows - 1) / kRows), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    h3d::pre_launch();
    if (vec_ok(C, {x})) hipLaunchKernelGGL((channel_moments<4, T>), grid, dim3(kThreads), 0, s, x, partial, P, C);
    else hipLaunchKernelGGL((channel_moments<1, T>), grid, dim3(kThreads), 0, s, x, partial, P, C);
    return h3d::launch_status("h3d_channel_moments");
}

template <typename T>
static int spade_fwd_any(const T* x, const float* scale, const float* shift, const void* gamma, const void* beta,
                         T* y, int B, int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream) {
    if (int rc = check_shape("h3d_spade_fwd", B, P, C)) return rc;
    if (B == 0 || P == 0) return H3D_OK;
    H3D_REQUIRE(x && scale && shift && gamma && beta && y, "h3d_spade_fwd: null pointer");
    const dim3 grid((unsigned)((P + kRows - 1) / kRows), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = vec_ok(C, {x, scale, shift, gamma, beta, y});
    h3d::pre_launch();
#define H3D_GO(V, PIX) hipLaunchKernelGGL((spade_fwd<V, PIX, T>), grid, dim3(kThreads), 0, s, x, scale, shift, \
        static_cast<const typename Mod<PIX, T>::type*>(gamma), static_cast<const typename Mod<PIX, T>::type*>(beta), y, P, C, slope)
    if (v4) { if (per_pixel) H3D_GO(4, true); else H3D_GO(4, false); }
    else { if (per_pixel) H3D_GO(1, true); else H3D_GO(1, false); }
#undef H3D_GO
    return h3d::launch_status("h3d_spade_fwd");
}

template <typename T>
static int spade_bwd_reduce_any(const T* x, const float* mean, const float* rstd, const float* g, const float* b,
                                const void* gamma, const void* beta, const T* dy, float* partial, int B, int64_t P,
                                int C, int per_pixel, float slope, h3d_stream_t stream) {
    if (int rc = check_shape("h3d_spade_bwd_reduce", B, P, C)) return rc;
    if (B == 0 || P == 0) return H3D_OK;
    H3D_REQUIRE(x && mean && rstd && g && b && gamma && beta && dy && partial, "h3d_spade_bwd_reduce: null pointer");
    const dim3 grid((unsigned)((P + kRows - 1) / kRows), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = vec_ok(C, {x, mean, rstd, g, b, gamma, beta, dy});
    h3d::pre_launch();
#define H3D_GO(V, PIX) hipLaunchKernelGGL((spade_bwd_reduce<V, PIX, T>), grid, dim3(kThreads), 0, s, x, mean, rstd, g, b, \
        static_cast<const typename Mod<PIX, T>::type*>(gamma), static_cast<const typename Mod<PIX, T>::type*>(beta), dy, partial, P, C, slope)
    if (v4) { if (per_pixel) H3D_GO(4, true); else H3D_GO(4, false); }
    else { if (per_pixel) H3D_GO(1, true); else H3D_GO(1, false); }
#undef H3D_GO
    return h3d::launch_status("h3d_spade_bwd_reduce");
}

template <typename T>
static int spade_bwd_apply_any(const T* x, const float* mean, const float* rstd, const float* g, const float* b,
                               const void* gamma, const void* beta, const T* dy, const float* c1, const float* c2,
                               T* dx, T* dgamma, T* dbeta, float* partial, int B, int64_t P, int C,
                               int per_pixel, float slope, h3d_stream_t stream, const T* add1 = nullptr, const T* add2 = nullptr) {
    if (int rc = check_shape("h3d_spade_bwd_apply", B, P, C)) return rc;
    if (B == 0 || P == 0) return H3D_OK;
    H3D_REQUIRE(x && mean && rstd && g && b && gamma && beta && dy && c1 && c2 && dx, "h3d_spade_bwd_apply: null pointer");
    H3D_REQUIRE(per_pixel ? (dgamma && dbeta) : (partial != nullptr),
                "h3d_spade_bwd_apply: %s", per_pixel ? "per-pixel mode needs dgamma and dbeta" : "per-sample mode needs partial");
    const dim3 grid((unsigned)((P + kRows - 1) / kRows), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = vec_ok(C, {x, mean, rstd, g, b, gamma, beta, dy, c1, c2, dx, dgamma, dbeta, add1, add2});
    h3d::pre_launch();
#define H3D_GO(V, PIX) hipLaunchKernelGGL((spade_bwd_apply<V, PIX, T>), grid, dim3(kThreads), 0, s, x, mean, rstd, g, b, \
        static_cast<const typename Mod<PIX, T>::type*>(gamma), static_cast<const typename Mod<PIX, T>::type*>(beta), dy, c1, c2, dx, dgamma, dbeta, partial, P, C, slope, add1, add2)
    if (v4) { if (per_pixel) H3D_GO(4, true); else H3D_GO(4, false); }
    else { if (per_pixel) H3D_GO(1, true); else H3D_GO(1, false); }
#undef H3D_GO
    return h3d::launch_status("h3d_spade_bwd_apply");
}

extern "C" int h3d_channel_moments(const float* x, float* partial, int B, int64_t P, int C, h3d_stream_t stream) {
    return channel_moments_any<float>(x, partial, B, P, C, stream);
}
extern "C" int h3d_spade_fwd(const float* x, const float* scale, const float* shift, const float* gamma, const float* beta,
                             float* y, int B, int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream) {
    return spade_fwd_any<float>(x, scale, shift, gamma, beta, y, B, P, C, per_pixel, slope, stream);
}
extern "C" int h3d_spade_bwd_reduce(const float* x, const float* mean, const float* rstd, const float* g, const float* b,
                                    const float* gamma, const float* beta, const float* dy, float* partial, int B, int64_t P,
                                    int C, int per_pixel, float slope, h3d_stream_t stream) {
    return spade_bwd_reduce_any<float>(x, mean, rstd, g, b, gamma, beta, dy, partial, B, P, C, per_pixel, slope, stream);
}
extern "C" int h3d_spade_bwd_apply(const float* x, const float* mean, const float* rstd, const float* g, const float* b,
                                   const float* gamma, const float* beta, const float* dy, const float* c1, const float* c2,
                                   float* dx, float* dgamma, float* dbeta, float* partial, int B, int64_t P, int C,
                                   int per_pixel, float slope, h3d_stream_t stream) {
    return spade_bwd_apply_any<float>(x, mean, rstd, g, b, gamma, beta, dy, c1, c2, dx, dgamma, dbeta, partial, B, P, C, per_pixel, slope, stream);
}

/* The same four passes on f16 activations (AMP tier, round 4): x, y, dy, dx and the PER-PIXEL gamma / beta / dgamma / dbeta are
 * _Float16; per-channel vectors, per-sample gamma / beta and every partial sum stay fp32; arithmetic in fp32 registers. */
extern "C" int h3d_channel_moments_f16(const void* x, float* partial, int B, int64_t P, int C, h3d_stream_t stream) {
    return channel_moments_any<_Float16>(static_cast<const _Float16*>(x), partial, B, P, C, stream);
}
extern "C" int h3d_spade_fwd_f16(const void* x, const float* scale, const float* shift, const void* gamma, const void* beta,
                                 void* y, int B, int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream) {
    return spade_fwd_any<_Float16>(static_cast<const _Float16*>(x), scale, shift, gamma, beta, static_cast<_Float16*>(y), B, P, C, per_pixel, slope, stream);
}
extern "C" int h3d_spade_bwd_reduce_f16(const void* x, const float* mean, const float* rstd, const float* g, const float* b,
                                        const void* gamma, const void* beta, const void* dy, float* partial, int B, int64_t P,
                                        int C, int per_pixel, float slope, h3d_stream_t stream) {
    return spade_bwd_reduce_any<_Float16>(static_cast<const _Float16*>(x), mean, rstd, g, b, gamma, beta, static_cast<const _Float16*>(dy), partial, B, P, C,
                                          per_pixel, slope, stream);
}
extern "C" int h3d_spade_bwd_apply_f16(const void* x, const float* mean, const float* rstd, const float* g, const float* b,
                                       const void* gamma, const void* beta, const void* dy, const float* c1, const float* c2,
                                       void* dx, void* dgamma, void* dbeta, float* partial, int B, int64_t P, int C,
                                       int per_pixel, float slope, h3d_stream_t stream) {
    return spade_bwd_apply_any<_Float16>(static_cast<const _Float16*>(x), mean, rstd, g, b, gamma, beta, static_cast<const _Float16*>(dy), c1, c2,
                                         static_cast<_Float16*>(dx), static_cast<_Float16*>(dgamma), static_cast<_Float16*>(dbeta), partial, B, P, C,
                                         per_pixel, slope, stream);
}

/* h3d_spade_bwd_apply[_f16] (dtype 0 / 1) whose dx also receives up to two more gradients of x (same type and shape as x, or NULL):
 * dx = (the SPADE term) + add1 + add2.  x of a skip block feeds the SPADE, the residual connection and the previous block's ToRGB
 * head (/root/reference/lib/components/map3d_layers.py:228-236, 268-272); autograd would add the three gradients in two passes of
 * its own (round 6). */
extern "C" int h3d_spade_bwd_apply_acc(int dtype, const void* x, const float* mean, const float* rstd, const float* g, const float* b,
                                       const void* gamma, const void* beta, const void* dy, const float* c1, const float* c2,
                                       const void* add1, const void* add2, void* dx, void* dgamma, void* dbeta, float* partial, int B,
                                       int64_t P, int C, int per_pixel, float slope, h3d_stream_t stream) {
    H3D_REQUIRE(dtype == 0 || dtype == 1, "h3d_spade_bwd_apply_acc: dtype %d (0 = f32, 1 = f16)", dtype);
    if (dtype == 0)
        return spade_bwd_apply_any<float>(static_cast<const float*>(x), mean, rstd, g, b, gamma, beta, static_cast<const float*>(dy), c1, c2,
                                          static_cast<float*>(dx), static_cast<float*>(dgamma), static_cast<float*>(dbeta), partial, B, P, C,
                                          per_pixel, slope, stream, static_cast<const float*>(add1), static_cast<const float*>(add2));
    return spade_bwd_apply_any<_Float16>(static_cast<const _Float16*>(x), mean, rstd, g, b, gamma, beta, static_cast<const _Float16*>(dy), c1, c2,
                                         static_cast<_Float16*>(dx), static_cast<_Float16*>(dgamma), static_cast<_Float16*>(dbeta), partial, B,
                                         P, C, per_pixel, slope, stream, static_cast<const _Float16*>(add1), static_cast<const _Float16*>(add2));
}

// ---------------------------------------------------------------- the bookkeeping between the passes (round 6)
// Around every SPADE the per-workgroup partial sums were finished by ~20 tensor operations on 256-element vectors (a double copy, a
// reduction, divisions, a clamp, two lerps, casts ...): 36 SPADE calls x 20 launches per config-4 iteration -- 12 % of its launches,
// ~3 ms of GPU time and, under AMP where the iteration is launch-bound, more than that of wall time.  Three small kernels instead:
//   h3d_rows_sum_f64    sums[c] = sum_r partial[r][c] in fp64 (what partial.double().sum(0) computed)
//   h3d_bn_finish       sums, count -> mean, rstd (fp32) and the running-statistics update of nn.BatchNorm (momentum lerp with the
//                       unbiased variance, num_batches_tracked += 1): /root/reference/lib/components/map3d_layers.py:162
//   h3d_bn_bwd_finish   backward sums, count -> d_bias, d_weight and the two batch-statistics terms c1, c2 (all fp32)
namespace {
__global__ __launch_bounds__(1024) void rows_sum_f64_kernel(const float* __restrict__ partial, double* __restrict__ out, int64_t n_rows,
                                                            int n_cols) {
    __shared__ double part[16][64];
    const int col = (int)blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < n_cols)
        for (int64_t r = grp; r < n_rows; r += 16) acc += (double)partial[r * n_cols + col];
    part[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && col < n_cols) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
        out[col] = t;
    }
}

// the same sums for n_cols % 4 == 0: a thread owns FOUR consecutive columns (one 16-byte load per row) and 64 row groups share a
// block -- h3d_conv_x3_moments hands over one row per 128 pixels (4 096 rows at config 4), where the 16-group kernel above spends
// 256 dependent iterations per thread (~100 us); here 64 iterations with four independent accumulators
__global__ __launch_bounds__(1024) void rows_sum_f64_v4_kernel(const float* __restrict__ partial, double* __restrict__ out, int64_t n_rows,
                                                               int n_cols) {
    __shared__ double part[64][65];
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int col = (int)blockIdx.x * 64 + 4 * q;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (col < n_cols) {
#pragma unroll 4
        for (int64_t r = grp; r < n_rows; r += 64) {
            const float4 v = *reinterpret_cast<const float4*>(partial + r * n_cols + col);
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
    }
    part[grp][4 * q] = a0; part[grp][4 * q + 1] = a1; part[grp][4 * q + 2] = a2; part[grp][4 * q + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 64 && (int)blockIdx.x * 64 + (int)threadIdx.x < n_cols) {
        double t = 0.0;
#pragma unroll 8
        for (int g = 0; g < 64; ++g) t += part[g][threadIdx.x];
        out[blockIdx.x * 64 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void bn_finish_kernel(const double* __restrict__ sums, const double* __restrict__ count,
                                                        float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var, int64_t* __restrict__ tracked, int C, float eps,
                                                        float momentum) {
    const int c = (int)blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && tracked) *tracked += 1;
    if (c >= C) return;
    const double n = count[0];
    const double m = sums[c] / n;
    double v = sums[C + c] / n - m * m;
    v = v > 0.0 ? v : 0.0;
    const float mf = (float)m, vf = (float)v;
    mean[c] = mf;
    rstd[c] = 1.0f / sqrtf(vf + eps);
    if (run_mean) run_mean[c] += momentum * (mf - run_mean[c]);
    if (run_var) {
        const double d = n - 1.0 > 1.0 ? n - 1.0 : 1.0;
        run_var[c] += momentum * ((float)(v * (n / d)) - run_var[c]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const double* __restrict__ local, const double* __restrict__ global,
                                                            const double* __restrict__ count, float* __restrict__ d_b,
                                                            float* __restrict__ d_g, float* __restrict__ c1, float* __restrict__ c2, int C) {
    const int c = (int)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    d_b[c] = (float)local[c];
    d_g[c] = (float)local[C + c];
    if (count) {
        const double n = count[0];
        c1[c] = (float)(global[c] / n);
        c2[c] = (float)(global[C + c] / n);
    } else {
        c1[c] = 0.f;
        c2[c] = 0.f;
    }
}
}  // namespace

extern "C" int h3d_rows_sum_f64(const float* partial, double* out, int64_t n_rows, int n_cols, h3d_stream_t stream) {
    H3D_REQUIRE(partial && out && n_rows >= 0 && n_cols >= 1, "h3d_rows_sum_f64: bad arguments");
    h3d::pre_launch();
    if (n_cols % 4 == 0 && h3d::aligned16(partial) && n_rows >= 256)
        hipLaunchKernelGGL(rows_sum_f64_v4_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                           partial, out, n_rows, n_cols);
    else
        hipLaunchKernelGGL(rows_sum_f64_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(1024), 0, static_cast<hipStream_t>(stream), partial,
                           out, n_rows, n_cols);
    return h3d::launch_status("h3d_rows_sum_f64");
}

extern "C" int h3d_bn_finish(const double* sums, const double* count, float* mean, float* rstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, int C, float eps, float momentum, h3d_stream_t stream) {
    H3D_REQUIRE(sums && count && mean && rstd && C >= 1, "h3d_bn_finish: bad arguments");
    h3d::pre_launch();
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sums, count, mean,
                       rstd, running_mean, running_var, num_batches_tracked, C, eps, momentum);
    return h3d::launch_status("h3d_bn_finish");
}

extern "C" int h3d_bn_bwd_finish(const double* local_sums, const double* global_sums, const double* count, float* d_bias, float* d_weight,
                                 float* c1, float* c2, int C, h3d_stream_t stream) {
    H3D_REQUIRE(local_sums && d_bias && d_weight && c1 && c2 && C >= 1, "h3d_bn_bwd_finish: bad arguments");
    H3D_REQUIRE((count == nullptr) || global_sums, "h3d_bn_bwd_finish: batch statistics need the (all-reduced) sums");
    h3d::pre_launch();
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), local_sums,
                       global_sums, count, d_bias, d_weight, c1, c2, C);
    return h3d::launch_status("h3d_bn_bwd_finish");
}

