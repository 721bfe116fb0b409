// The discriminator's resampling and activation glue for gfx950, fused (reference: lib/discriminators/unet_discriminators.py:8-72 --
// nn.Sequential(LeakyReLU(0.2), Upsample(scale_factor=2), conv) in the up blocks, AvgPool2d(2) and the residual sum in forward()).
// Two HBM-bound kernels on channels-last activations [B, H, W, C] (fp32, or f16 in the AMP tier; arithmetic in fp32 registers):
//
//   h3d_up2_mask    out[b, 2y+i, 2x+j, c] = scale * m(mask[b,y,x,c]) * x[b,y,x,c] (+ addend[b, 2y+i, 2x+j, c])
//   h3d_pool2_mask  out[b, y, x, c]       = scale * m(mask[b,y,x,c]) * sum_ij (x (+ x2))[b, 2y+i, 2x+j, c]
//
// with m(t) = 1 for t > 0, `slope` otherwise (the derivative of LeakyReLU; m = 1 without a mask tensor).  For a fixed mask the two
// are adjoint linear maps, so each is the other's backward and autograd can differentiate the pair to any order (the R1 penalty
// differentiates the discriminator's backward pass):
//   up(lrelu(x))        = up2_mask(x, mask = x)            one pass instead of two, no lrelu(x) tensor
//   its backward        = pool2_mask(g, mask = x)          instead of upsample_backward + leaky_relu_backward
//   avgpool(s + d)      = pool2_mask(s, x2 = d, 1/4)       no full-resolution sum tensor
//   up(s) + d           = up2_mask(s, addend = d)
// A thread owns V consecutive channels (16 bytes) of one low-resolution pixel: consecutive lanes touch consecutive 16 bytes.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

template <typename T, int V> struct Vec { typedef T type __attribute__((ext_vector_type(V))); };

template <typename T, int V> __device__ __forceinline__ void load(const T* p, float (&v)[V]) {
    if constexpr (V == 1) {
        v[0] = (float)p[0];
    } else {
        const typename Vec<T, V>::type t = *reinterpret_cast<const typename Vec<T, V>::type*>(p);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = (float)t[k];
    }
}
template <typename T, int V> __device__ __forceinline__ void store(T* p, const float (&v)[V]) {
    if constexpr (V == 1) {
        p[0] = (T)v[0];
    } else {
        typename Vec<T, V>::type t;
#pragma unroll
        for (int k = 0; k < V; ++k) t[k] = (T)v[k];
        *reinterpret_cast<typename Vec<T, V>::type*>(p) = t;
    }
}

template <typename T, int V, bool MASK, bool ADD>
__global__ __launch_bounds__(kThreads) void up2_kernel(const T* __restrict__ x, const T* __restrict__ mask, const T* __restrict__ addend,
                                                       T* __restrict__ out, int64_t n_items, int H, int W, int C, float slope, float scale) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_items) return;
    const int Q = C / V;
    const int q = (int)(i % Q);
    const int64_t pix = i / Q;                       // (b, y, x) row-major
    const int xx = (int)(pix % W);
    const int64_t by = pix / W;                      // b * H + y
    float v[V];
    load<T, V>(x + pix * C + q * V, v);
    if constexpr (MASK) {
        float m[V];
        load<T, V>(mask + pix * C + q * V, m);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] *= m[k] > 0.f ? scale : scale * slope;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] *= scale;
    }
    const int64_t row = (int64_t)2 * W * C;          // one output row
    const int64_t o00 = (by * 2) * row + (int64_t)(2 * xx) * C + q * V;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int64_t o = o00 + dy * row + dx * C;
            if constexpr (ADD) {
                float a[V], r[V];
                load<T, V>(addend + o, a);
#pragma unroll
                for (int k = 0; k < V; ++k) r[k] = v[k] + a[k];
                store<T, V>(out + o, r);
            } else {
                store<T, V>(out + o, v);
            }
        }
}

template <typename T, int V, bool MASK, bool TWO>
__global__ __launch_bounds__(kThreads) void pool2_kernel(const T* __restrict__ x, const T* __restrict__ x2, const T* __restrict__ mask,
                                                         T* __restrict__ out, int64_t n_items, int Ho, int Wo, int C, float slope, float scale) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_items) return;
    const int Q = C / V;
    const int q = (int)(i % Q);
    const int64_t pix = i / Q;                       // output (b, y, x)
    const int xx = (int)(pix % Wo);
    const int64_t by = pix / Wo;
    const int64_t row = (int64_t)2 * Wo * C;
    const int64_t i00 = (by * 2) * row + (int64_t)(2 * xx) * C + q * V;
    float s[2][V];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        float a[V], b[V];
        load<T, V>(x + i00 + dy * row, a);
        load<T, V>(x + i00 + dy * row + C, b);
        if constexpr (TWO) {
            float c[V], d[V];
            load<T, V>(x2 + i00 + dy * row, c);
            load<T, V>(x2 + i00 + dy * row + C, d);
#pragma unroll
            for (int k = 0; k < V; ++k) { a[k] += c[k]; b[k] += d[k]; }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) s[dy][k] = a[k] + b[k];
    }
    float r[V];
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = s[0][k] + s[1][k];
    if constexpr (MASK) {
        float m[V];
        load<T, V>(mask + pix * C + q * V, m);
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] *= m[k] > 0.f ? scale : scale * slope;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] *= scale;
    }
    store<T, V>(out + pix * C + q * V, r);
}

template <typename T, int V>
int launch_up2(const void* x, const void* mask, const void* addend, void* out, int B, int H, int W, int C, float slope, float scale,
               hipStream_t st) {
    const int64_t n = (int64_t)B * H * W * (C / V);
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads)), block(kThreads);
    const T *xp = static_cast<const T*>(x), *mp = static_cast<const T*>(mask), *ap = static_cast<const T*>(addend);
    T* op = static_cast<T*>(out);
    h3d::pre_launch();
    if (mask && addend) hipLaunchKernelGGL((up2_kernel<T, V, true, true>), grid, block, 0, st, xp, mp, ap, op, n, H, W, C, slope, scale);
    else if (mask) hipLaunchKernelGGL((up2_kernel<T, V, true, false>), grid, block, 0, st, xp, mp, ap, op, n, H, W, C, slope, scale);
    else if (addend) hipLaunchKernelGGL((up2_kernel<T, V, false, true>), grid, block, 0, st, xp, mp, ap, op, n, H, W, C, slope, scale);
    else hipLaunchKernelGGL((up2_kernel<T, V, false, false>), grid, block, 0, st, xp, mp, ap, op, n, H, W, C, slope, scale);
    return h3d::launch_status("h3d_up2_mask");
}

template <typename T, int V>
int launch_pool2(const void* x, const void* x2, const void* mask, void* out, int B, int Ho, int Wo, int C, float slope, float scale,
                 hipStream_t st) {
    const int64_t n = (int64_t)B * Ho * Wo * (C / V);
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads)), block(kThreads);
    const T *xp = static_cast<const T*>(x), *yp = static_cast<const T*>(x2), *mp = static_cast<const T*>(mask);
    T* op = static_cast<T*>(out);
    h3d::pre_launch();
    if (mask && x2) hipLaunchKernelGGL((pool2_kernel<T, V, true, true>), grid, block, 0, st, xp, yp, mp, op, n, Ho, Wo, C, slope, scale);
    else if (mask) hipLaunchKernelGGL((pool2_kernel<T, V, true, false>), grid, block, 0, st, xp, yp, mp, op, n, Ho, Wo, C, slope, scale);
    else if (x2) hipLaunchKernelGGL((pool2_kernel<T, V, false, true>), grid, block, 0, st, xp, yp, mp, op, n, Ho, Wo, C, slope, scale);
    else hipLaunchKernelGGL((pool2_kernel<T, V, false, false>), grid, block, 0, st, xp, yp, mp, op, n, Ho, Wo, C, slope, scale);
    return h3d::launch_status("h3d_pool2_mask");
}

bool all16(const void* a, const void* b, const void* c, const void* d) {
    return h3d::aligned16(a) && h3d::aligned16(b) && h3d::aligned16(c) && h3d::aligned16(d);      // null is aligned
}

}  // namespace

extern "C" int h3d_up2_mask(const void* x, const void* mask, const void* addend, void* out, int B, int H, int W, int C, float slope,
                            float scale, int half, h3d_stream_t stream) {
    H3D_REQUIRE(x && out, "h3d_up2_mask: null pointer");
    H3D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 1, "h3d_up2_mask: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    H3D_REQUIRE((int64_t)B * H * W * C < (int64_t(1) << 38), "h3d_up2_mask: tensor too large");
    if (B == 0) return H3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = all16(x, mask, addend, out);
    if (half) {
        if (vec && C % 8 == 0) return launch_up2<_Float16, 8>(x, mask, addend, out, B, H, W, C, slope, scale, st);
        return launch_up2<_Float16, 1>(x, mask, addend, out, B, H, W, C, slope, scale, st);
    }
    if (vec && C % 4 == 0) return launch_up2<float, 4>(x, mask, addend, out, B, H, W, C, slope, scale, st);
    return launch_up2<float, 1>(x, mask, addend, out, B, H, W, C, slope, scale, st);
}

extern "C" int h3d_pool2_mask(const void* x, const void* x2, const void* mask, void* out, int B, int Ho, int Wo, int C, float slope,
                              float scale, int half, h3d_stream_t stream) {
    H3D_REQUIRE(x && out, "h3d_pool2_mask: null pointer");
    H3D_REQUIRE(B >= 0 && Ho >= 1 && Wo >= 1 && C >= 1, "h3d_pool2_mask: bad shape B=%d Ho=%d Wo=%d C=%d", B, Ho, Wo, C);
    H3D_REQUIRE((int64_t)B * Ho * Wo * C < (int64_t(1) << 36), "h3d_pool2_mask: tensor too large");
    if (B == 0) return H3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = all16(x, x2, mask, out);
    if (half) {
        if (vec && C % 8 == 0) return launch_pool2<_Float16, 8>(x, x2, mask, out, B, Ho, Wo, C, slope, scale, st);
        return launch_pool2<_Float16, 1>(x, x2, mask, out, B, Ho, Wo, C, slope, scale, st);
    }
    if (vec && C % 4 == 0) return launch_pool2<float, 4>(x, x2, mask, out, B, Ho, Wo, C, slope, scale, st);
    return launch_pool2<float, 1>(x, x2, mask, out, B, Ho, Wo, C, slope, scale, st);
}
