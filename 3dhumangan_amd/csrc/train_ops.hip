// Training-side streaming kernels for gfx950 (backward passes of A5 / A6).  All HBM-bound.
//
//   h3d_film_sin / h3d_film_sin_bwd      y = sin(freq[b,c] * x[b,n,c] + phase[b,c])  -- the activation of SineLayer / FiLMLayer
//                                        (lib/components/pigan_layers.py:63-87) fused into one pass each way; the backward
//                                        recomputes the cosine instead of keeping three intermediates alive.
//   h3d_ray_integrate_bwd                gradient of lib/generators/volume_rendering.py:12-56 (ray_integration) w.r.t. the
//                                        field tensor: one read of the field, one write of its gradient.
#include "common.hpp"
#include <hip/hip_fp16.h>

namespace {

constexpr int kThreads = 256;

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec4<__half> {
    static __device__ __forceinline__ void load(const __half* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        const __half2 a = *reinterpret_cast<const __half2*>(&t.x), b = *reinterpret_cast<const __half2*>(&t.y);
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    }
    static __device__ __forceinline__ void store(__half* p, const float (&v)[4]) {
        const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
        uint2 t;
        t.x = *reinterpret_cast<const unsigned*>(&a);
        t.y = *reinterpret_cast<const unsigned*>(&b);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

template <typename T, int V> __device__ __forceinline__ void load_v(const T* p, float (&v)[V]) {
    if constexpr (V == 4) Vec4<T>::load(p, v);
    else v[0] = (float)p[0];
}
template <typename T, int V> __device__ __forceinline__ void store_v(T* p, const float (&v)[V]) {
    if constexpr (V == 4) Vec4<T>::store(p, v);
    else p[0] = (T)v[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// film_sin.  x, y [B, N, C]; freq, phase [B, C] fp32 or NULL (then y = sin(w0 * x)).  A workgroup owns `rows` consecutive
// rows of one batch item; thread t owns V consecutive channels (quad q = t % QP) of the rows g, g + G, ... (g = t / QP),
// so its frequency / phase values stay in registers and consecutive lanes touch consecutive 16 bytes.
template <typename T, int V>
__global__ __launch_bounds__(kThreads) void film_sin_fwd(const T* __restrict__ x, const float* __restrict__ freq,
                                                         const float* __restrict__ phase, T* __restrict__ y, int64_t N,
                                                         int C, int rows, float w0) {
    const int Q = C / V, QP = Q < kThreads ? Q : kThreads, G = kThreads / QP;
    const int t = threadIdx.x, g = t / QP;
    if (g >= G) return;
    const int b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows;
    const int64_t r1 = r0 + rows < N ? r0 + rows : N;
    for (int q = t - g * QP; q < Q; q += QP) {
        float f[V], p[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            f[k] = freq ? freq[(int64_t)b * C + q * V + k] : w0;
            p[k] = phase ? phase[(int64_t)b * C + q * V + k] : 0.f;
        }
        for (int64_t r = r0 + g; r < r1; r += G) {
            const int64_t off = ((int64_t)b * N + r) * C + q * V;
            float v[V];
            load_v<T, V>(x + off, v);
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = sinf(fmaf(f[k], v[k], p[k]));
            store_v<T, V>(y + off, v);
        }
    }
}

// t = dy * cos(freq * x + phase);  dx = t * freq;  partial[b][blk][0][c] = sum_rows t * x,  partial[b][blk][1][c] = sum_rows t
// (deterministic two-stage reduction: the caller sums `partial` over blk).
template <typename T, int V>
__global__ __launch_bounds__(kThreads) void film_sin_bwd(const T* __restrict__ x, const float* __restrict__ freq,
                                                         const float* __restrict__ phase, const T* __restrict__ dy,
                                                         T* __restrict__ dx, float* __restrict__ partial, int64_t N, int C,
                                                         int rows, float w0) {
    __shared__ float red[2][kThreads][V];
    const int Q = C / V, QP = Q < kThreads ? Q : kThreads, G = kThreads / QP;
    const int t = threadIdx.x, g = t / QP;
    const int b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows;
    const int64_t r1 = r0 + rows < N ? r0 + rows : N;
    float* out = partial ? partial + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C : nullptr;
    for (int q0 = 0; q0 < Q; q0 += QP) {
        const int q = q0 + t - g * QP;
        const bool on = g < G && q < Q;
        float sf[V], sp[V];
#pragma unroll
        for (int k = 0; k < V; ++k) sf[k] = sp[k] = 0.f;
        if (on) {
            float f[V], p[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                f[k] = freq ? freq[(int64_t)b * C + q * V + k] : w0;
                p[k] = phase ? phase[(int64_t)b * C + q * V + k] : 0.f;
            }
            for (int64_t r = r0 + g; r < r1; r += G) {
                const int64_t off = ((int64_t)b * N + r) * C + q * V;
                float v[V], d[V];
                load_v<T, V>(x + off, v);
                load_v<T, V>(dy + off, d);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float tt = d[k] * cosf(fmaf(f[k], v[k], p[k]));
                    sf[k] = fmaf(tt, v[k], sf[k]);
                    sp[k] += tt;
                    d[k] = tt * f[k];
                }
                store_v<T, V>(dx + off, d);
            }
        }
        if (out) {                                          // uniform branch: every thread reaches the barriers
#pragma unroll
            for (int k = 0; k < V; ++k) { red[0][t][k] = sf[k]; red[1][t][k] = sp[k]; }
            __syncthreads();
            if (g == 0 && q < Q) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float a = 0.f, c = 0.f;
                    for (int gg = 0; gg < G; ++gg) { a += red[0][gg * QP + t][k]; c += red[1][gg * QP + t][k]; }
                    out[q * V + k] = a;
                    out[C + q * V + k] = c;
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ray_integrate backward.  One workgroup per ray.
//   1 (wave 0)   forward scan again: alpha, transmittance T, weights w (pre-last_back), d(alpha)/d(sigma) -> LDS
//   2 (4 waves)  a_s = sum_c gO[c] * F[s][c]  (row s by wave s % 4; lanes stride the row 16 bytes at a time)
//   3 (wave 0)   g_s = dL/dw_s, suffix sums, dL/dsigma_s -> LDS
//   4 (4 waves)  dF[s][c] = w'_s * gO[c], dF[s][C] = dL/dsigma_s
__device__ __forceinline__ float density_fn(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? x : log1pf(expf(x));
    return fmaxf(x, 0.f);
}
__device__ __forceinline__ float density_deriv(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
    return x > 0.f ? 1.f : 0.f;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int V>
__global__ __launch_bounds__(kThreads) void ray_integrate_bwd_kernel(
    const float* __restrict__ field, const float* __restrict__ z_vals, const float* __restrict__ noise,
    const float* __restrict__ g_feats, const float* __restrict__ g_depth, const float* __restrict__ g_weights,
    float* __restrict__ d_field, int S, int C, int clamp_mode, int last_back, int white_back) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w_s = smem;                 // pre-last_back weights
    float* T_s = smem + S;             // transmittance in front of sample s
    float* f_s = smem + 2 * S;         // 1 - alpha + 1e-12
    float* da_s = smem + 3 * S;        // d alpha / d sigma
    float* a_s = smem + 4 * S;         // phase 2: sum_c gO F;  phase 3 overwrites it with dL/dsigma
    float* misc = smem + 5 * S;        // [0] = background term 1 - sum w
    const int64_t ray = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int W = C + 1;
    const float* __restrict__ blk = field + ray * (int64_t)S * W;
    const float* __restrict__ z = z_vals + ray * S;
    const float* __restrict__ nz = noise ? noise + ray * S : nullptr;
    const float* __restrict__ gO = g_feats + ray * (int64_t)C;

    if (wave == 0) {
        float carry = 1.f, wsum = 0.f;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool ok = s < S;
            float f = 1.f, alpha = 0.f, da = 0.f;
            if (ok) {
                float sg = blk[(int64_t)s * W + C];
                if (nz) sg += nz[s];
                const float delta = (s == S - 1) ? 1e9f : z[s + 1] - z[s];
                const float e = expf(-delta * density_fn(sg, clamp_mode));          // = 1 - alpha
                alpha = 1.f - e;
                f = (1.f - alpha) + 1e-12f;
                da = delta * e * density_deriv(sg, clamp_mode);
            }
            float incl = f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float u = __shfl_up(incl, off, 64);
                if (lane >= off) incl *= u;
            }
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            const float T = carry * excl;
            carry *= __shfl(incl, 63, 64);
            if (ok) { w_s[s] = alpha * T; T_s[s] = T; f_s[s] = f; da_s[s] = da; }
            wsum += alpha * T;
        }
        wsum = wave_sum(wsum);
        if (lane == 0) misc[0] = 1.f - wsum;
    }

    // the upstream gradient of this ray's C output channels, V at a time per lane; element C (the density slot) = 0
    constexpr int kMaxIt = 8;                                   // 64 lanes * V * 8 >= C + 1
    const int n_it = (W + 64 * V - 1) / (64 * V);
    float go[kMaxIt][V];
    float go_sum = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = (it * 64 + lane) * V + k;
            go[it][k] = (it < n_it && c < C) ? gO[c] : 0.f;
            go_sum += go[it][k];
        }
    }
    go_sum = wave_sum(go_sum);

    for (int s = wave; s < S; s += 4) {
        const float* __restrict__ row = blk + (int64_t)s * W;
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            if (it < n_it) {
                const int c0 = (it * 64 + lane) * V;
                if (c0 < W) {
                    float v[V];
                    if constexpr (V == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(row + c0);
                        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                    } else {
                        v[0] = row[c0];
                    }
#pragma unroll
                    for (int k = 0; k < V; ++k) acc = fmaf(go[it][k], v[k], acc);
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) a_s[s] = acc;
    }
    __syncthreads();

    if (wave == 0) {
        const float bg = misc[0];
        const float z_last = z[S - 1];
        const float gD = g_depth ? g_depth[ray] : 0.f;
        const float* __restrict__ gW = g_weights ? g_weights + ray * S : nullptr;
        const float a_last = a_s[S - 1] + (gW ? gW[S - 1] : 0.f);
        float carry = 0.f;                                         // sum_{k > s} g_k w_k, built from the far end
        const int n_chunks = (S + 63) / 64;
        for (int ch = n_chunks - 1; ch >= 0; --ch) {
            const int s = ch * 64 + lane;
            const bool ok = s < S;
            float g = 0.f, gw = 0.f;
            if (ok) {
                g = a_s[s] + (gW ? gW[s] : 0.f);
                if (last_back) g -= a_last;
                if (white_back) g -= go_sum;
                g += gD * (z[s] - z_last);
                gw = g * w_s[s];
            }
            float incl = gw;                                       // inclusive suffix sum within the chunk
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float u = __shfl_down(incl, off, 64);
                if (lane + off < 64) incl += u;
            }
            const float after = carry + incl - gw;                 // strictly behind s
            carry += __shfl(incl, 0, 64);
            if (ok) a_s[s] = (g * T_s[s] - after / f_s[s]) * da_s[s];
        }
        (void)bg;
    }
    __syncthreads();

    const float bg = misc[0];
    for (int s = wave; s < S; s += 4) {
        float* __restrict__ row = d_field + (ray * (int64_t)S + s) * W;
        const float w = w_s[s] + ((last_back && s == S - 1) ? bg : 0.f);
        const float dsig = a_s[s];
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            if (it < n_it) {
                const int c0 = (it * 64 + lane) * V;
                if (c0 < W) {
                    float v[V];
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] = (c0 + k == C) ? dsig : w * go[it][k];
                    if constexpr (V == 4) *reinterpret_cast<float4*>(row + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    else row[c0] = v[0];
                }
            }
        }
    }
}

template <typename T>
int launch_film(bool bwd, const void* x, const float* freq, const float* phase, const void* dy, void* out, float* partial,
                int B, int64_t N, int C, int rows, float w0, hipStream_t st) {
    const bool v4 = (C % 4 == 0) && h3d::aligned16(x) && h3d::aligned16(out) && (!dy || h3d::aligned16(dy));
    const dim3 grid((unsigned)((N + rows - 1) / rows), (unsigned)B);
    h3d::pre_launch();
    if (!bwd) {
        if (v4) hipLaunchKernelGGL((film_sin_fwd<T, 4>), grid, dim3(kThreads), 0, st, (const T*)x, freq, phase, (T*)out, N, C, rows, w0);
        else hipLaunchKernelGGL((film_sin_fwd<T, 1>), grid, dim3(kThreads), 0, st, (const T*)x, freq, phase, (T*)out, N, C, rows, w0);
    } else {
        if (v4) hipLaunchKernelGGL((film_sin_bwd<T, 4>), grid, dim3(kThreads), 0, st, (const T*)x, freq, phase, (const T*)dy, (T*)out, partial, N, C, rows, w0);
        else hipLaunchKernelGGL((film_sin_bwd<T, 1>), grid, dim3(kThreads), 0, st, (const T*)x, freq, phase, (const T*)dy, (T*)out, partial, N, C, rows, w0);
    }
    return h3d::launch_status(bwd ? "h3d_film_sin_bwd" : "h3d_film_sin");
}

}  // namespace

extern "C" int h3d_film_sin_rows(void) { return 512; }

extern "C" int h3d_film_sin(const void* x, const float* freq, const float* phase, void* y, int B, int64_t N, int C,
                            int dtype, float w0, h3d_stream_t stream) {
    H3D_REQUIRE(B >= 0 && N >= 0 && C >= 1, "h3d_film_sin: bad shape B=%d N=%lld C=%d", B, (long long)N, C);
    if (B == 0 || N == 0) return H3D_OK;
    H3D_REQUIRE(x && y, "h3d_film_sin: null pointer");
    H3D_REQUIRE((freq == nullptr) == (phase == nullptr), "h3d_film_sin: freq and phase must both be given or both be NULL");
    H3D_REQUIRE(dtype == 0 || dtype == 1, "h3d_film_sin: dtype %d (0 = f32, 1 = f16)", dtype);
    H3D_REQUIRE(B <= 65535, "h3d_film_sin: B=%d > 65535", B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = h3d_film_sin_rows();
    return dtype == 0 ? launch_film<float>(false, x, freq, phase, nullptr, y, nullptr, B, N, C, rows, w0, st)
                      : launch_film<__half>(false, x, freq, phase, nullptr, y, nullptr, B, N, C, rows, w0, st);
}

extern "C" int h3d_film_sin_bwd(const void* x, const float* freq, const float* phase, const void* dy, void* dx,
                                float* partial, int B, int64_t N, int C, int dtype, float w0, h3d_stream_t stream) {
    H3D_REQUIRE(B >= 0 && N >= 0 && C >= 1, "h3d_film_sin_bwd: bad shape B=%d N=%lld C=%d", B, (long long)N, C);
    if (B == 0 || N == 0) return H3D_OK;
    H3D_REQUIRE(x && dy && dx, "h3d_film_sin_bwd: null pointer");
    H3D_REQUIRE((freq == nullptr) == (phase == nullptr), "h3d_film_sin_bwd: freq and phase must both be given or both be NULL");
    H3D_REQUIRE(dtype == 0 || dtype == 1, "h3d_film_sin_bwd: dtype %d (0 = f32, 1 = f16)", dtype);
    H3D_REQUIRE(B <= 65535, "h3d_film_sin_bwd: B=%d > 65535", B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = h3d_film_sin_rows();
    return dtype == 0 ? launch_film<float>(true, x, freq, phase, dy, dx, partial, B, N, C, rows, w0, st)
                      : launch_film<__half>(true, x, freq, phase, dy, dx, partial, B, N, C, rows, w0, st);
}

extern "C" int h3d_ray_integrate_bwd(const float* field, const float* z_vals, const float* noise, const float* g_feats,
                                     const float* g_depth, const float* g_weights, float* d_field, int64_t n_rays, int S,
                                     int C, int clamp_mode, int last_back, int white_back, h3d_stream_t stream) {
    H3D_REQUIRE(field && z_vals && g_feats && d_field, "h3d_ray_integrate_bwd: null pointer");
    H3D_REQUIRE(n_rays >= 0 && n_rays < (int64_t(1) << 31), "h3d_ray_integrate_bwd: n_rays=%lld out of range", (long long)n_rays);
    H3D_REQUIRE(S >= 1 && S <= 2048, "h3d_ray_integrate_bwd: S=%d must be in [1,2048]", S);
    H3D_REQUIRE(C >= 1 && C + 1 <= 2048, "h3d_ray_integrate_bwd: C=%d must be in [1,2047]", C);
    H3D_REQUIRE(clamp_mode == 0 || clamp_mode == 1, "h3d_ray_integrate_bwd: clamp_mode must be 0 (relu) or 1 (softplus)");
    if (n_rays == 0) return H3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = sizeof(float) * (5 * (size_t)S + 4);
    const bool v4 = ((C + 1) % 4 == 0) && h3d::aligned16(field) && h3d::aligned16(d_field);
    h3d::pre_launch();
    if (v4)
        hipLaunchKernelGGL(ray_integrate_bwd_kernel<4>, dim3((unsigned)n_rays), dim3(kThreads), lds, st, field, z_vals, noise,
                           g_feats, g_depth, g_weights, d_field, S, C, clamp_mode, last_back, white_back);
    else {
        H3D_REQUIRE(C + 1 <= 512, "h3d_ray_integrate_bwd: C+1=%d not a multiple of 4 must be <= 512", C + 1);
        hipLaunchKernelGGL(ray_integrate_bwd_kernel<1>, dim3((unsigned)n_rays), dim3(kThreads), lds, st, field, z_vals, noise,
                           g_feats, g_depth, g_weights, d_field, S, C, clamp_mode, last_back, white_back);
    }
    return h3d::launch_status("h3d_ray_integrate_bwd");
}

// ---------------------------------------------------------------- zero-padded channels (round 6)
// out[b, p, 0 .. Cout) (channels-last, contiguous) = in[b, c, p] for c < Cin, 0 beyond: what lib/components/ops/conv.py needs in
// front of a convolution whose channel count is not a multiple of 64 (the discriminator's RGB stem,
// /root/reference/lib/discriminators/unet_discriminators.py:117) and behind the gradient of one whose output was narrowed (the
// 1- and label_dim-channel heads, :145-146).  One pass and one write of the padded tensor, from any input layout (element strides
// sb, sc, sp for batch, channel, pixel) -- torch built it as a zero fill, a layout-changing copy of the zeros and a channel
// concatenation whose result could come out NCHW and was copied once more (~0.7 ms per call at 4 x 64 x 512 x 256).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void pad_channels_kernel(const T* __restrict__ in, T* __restrict__ out, int Cin, int Cout, int64_t HW,
                                                           int64_t sb, int64_t sc, int64_t sp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread: 4 consecutive output channels of a pixel
    if (idx >= total) return;
    const int groups = Cout / 4;
    const int g = (int)(idx % groups);
    const int64_t pix = idx / groups, b = pix / HW, p = pix - b * HW;
    const T* src = in + b * sb + p * sp;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j;
        v[j] = c < Cin ? (float)src[c * sc] : 0.f;
    }
    Vec4<T>::store(out + pix * Cout + 4 * g, v);
}
}  // namespace

extern "C" int h3d_pad_channels_cl(const void* in, void* out, int B, int Cin, int Cout, int64_t HW, int64_t sb, int64_t sc, int64_t sp,
                                   int dtype, h3d_stream_t stream) {
    H3D_REQUIRE(B >= 0 && HW >= 0 && Cin >= 1 && Cout >= Cin && Cout % 4 == 0, "h3d_pad_channels_cl: bad shape B=%d Cin=%d Cout=%d", B, Cin, Cout);
    if (B == 0 || HW == 0) return H3D_OK;
    H3D_REQUIRE(in && out && h3d::aligned16(out), "h3d_pad_channels_cl: null input or an output that is not 16-byte aligned");
    H3D_REQUIRE(dtype == 0 || dtype == 1, "h3d_pad_channels_cl: dtype %d (0 = f32, 1 = f16)", dtype);
    const int64_t total = (int64_t)B * HW * (Cout / 4);
    H3D_REQUIRE((total + 255) / 256 <= 0x7fffffff, "h3d_pad_channels_cl: too many elements");
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    h3d::pre_launch();
    if (dtype == 0) hipLaunchKernelGGL(pad_channels_kernel<float>, grid, dim3(256), 0, st, (const float*)in, (float*)out, Cin, Cout, HW, sb, sc, sp, total);
    else hipLaunchKernelGGL(pad_channels_kernel<__half>, grid, dim3(256), 0, st, (const __half*)in, (__half*)out, Cin, Cout, HW, sb, sc, sp, total);
    return h3d::launch_status("h3d_pad_channels_cl");
}
