// P1 fused bias + activation (+gain, +clamp) for gfx950: forward and the two gradient orders.
// Replaces _plugin.bias_act (lib/components/ops/bias_act.cpp:32; grad = 0 -> h3d_bias_act, grad = 1 / 2 ->
// h3d_bias_act_grad); per-element formulas follow the kernel spec lib/components/ops/bias_act.cu:23-147:
//   y = clamp(act(x + b[(i/stepB) % sizeB]) * gain).
// Pure HBM streaming: 2 * n * sizeof(T) bytes forward.  fp32 goes through 16-byte accesses when alignment allows.
#include "common.hpp"
#include <hip/hip_fp16.h>

namespace {

template <typename A>   // A = float (f16/f32 compute in fp32, as the reference kernel does) or double
__device__ __forceinline__ A activate(A x, int act, A alpha) {
    switch (act) {
        default:
        case 1: return x;
        case 2: return x > 0 ? x : A(0);
        case 3: return x > 0 ? x : x * alpha;
        case 4: return tanh(x);
        case 5: return A(1) / (A(1) + exp(-x));
        case 6: return x >= 0 ? x : exp(x) - A(1);
        case 7: return x >= 0 ? x * A(1.0507009873554804934193349852946)
                              : (exp(x) - A(1)) * A(1.7580993408473768599402175208123);
        case 8: return x > A(20) ? x : log1p(exp(x));     // torch softplus threshold
        case 9: return x / (A(1) + exp(-x));
    }
}

template <typename T> struct Acc { using type = float; };
template <> struct Acc<double> { using type = double; };

template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y,
                                                       int64_t n, int64_t size_b, int64_t step_b, int act, float alpha,
                                                       float gain, float clamp) {
    using A = typename Acc<T>::type;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        A v = (A)x[i];
        if (b) v += (A)b[(i / step_b) % size_b];
        v = activate<A>(v, act, (A)alpha) * (A)gain;
        if (clamp >= 0.f) v = v > (A)clamp ? (A)clamp : (v < -(A)clamp ? -(A)clamp : v);
        y[i] = (T)v;
    }
}

// fp32, n % 4 == 0, 16-byte aligned, bias constant over each float4 (step_b % 4 == 0) or absent.
// NT: streaming (non-temporal) stores for outputs larger than the caches (round 6: 4.79 -> 5.34 TB/s on the [8,256,512,256] shape
// of the roofline table, same lease; a template parameter -- a run-time branch between the store kinds is merged into a plain store).
template <bool NT>
__global__ __launch_bounds__(256) void bias_act_f32x4(const float4* __restrict__ x, const float* __restrict__ b,
                                                      float4* __restrict__ y, int64_t n4, int64_t size_b, int64_t step_b,
                                                      int act, float alpha, float gain, float clamp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        const float bb = b ? b[((i * 4) / step_b) % size_b] : 0.f;
        float r[4] = {v.x + bb, v.y + bb, v.z + bb, v.w + bb};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = activate<float>(r[k], act, alpha) * gain;
            if (clamp >= 0.f) a = fminf(fmaxf(a, -clamp), clamp);
            r[k] = a;
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        if constexpr (NT) __builtin_nontemporal_store(f4{r[0], r[1], r[2], r[3]}, reinterpret_cast<f4*>(y + i));
        else y[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// First (order 1) or second (order 2) derivative of the activation, written in terms of what the forward pass kept:
// yy = y / gain for the activations whose derivative is a function of the output, xr = x + b for swish.
template <typename A>
__device__ __forceinline__ A activate_deriv(int order, int act, A yy, A xr, A alpha) {
    const A one = A(1), two = A(2);
    const A selu_s = A(1.0507009873554804934193349852946), selu_sa = A(1.7580993408473768599402175208123);
    if (order == 1) {
        switch (act) {
            default:
            case 1: return one;
            case 2: return yy > 0 ? one : A(0);
            case 3: return yy > 0 ? one : alpha;
            case 4: return one - yy * yy;
            case 5: return yy * (one - yy);
            case 6: return yy >= 0 ? one : yy + one;
            case 7: return yy >= 0 ? selu_s : yy + selu_sa;
            case 8: return one - exp(-yy);
            case 9: {
                if (xr > A(40)) return one;
                const A c = exp(xr), d = c + one;
                return c * (xr + d) / (d * d);
            }
        }
    }
    switch (act) {
        default: return A(0);                                   // piecewise-linear activations
        case 4: return (one - yy * yy) * (-two * yy);
        case 5: return yy * (one - yy) * (one - two * yy);
        case 6: return yy >= 0 ? A(0) : yy + one;
        case 7: return yy >= 0 ? A(0) : yy + selu_sa;
        case 8: { const A c = exp(-yy); return c * (one - c); }
        case 9: {
            if (xr > A(40)) return A(0);
            const A c = exp(xr), d = c + one;
            return c * (xr * (two - d) + two * d) / (d * d * d);
        }
    }
}

// out = g * [dy2] * gain * act^(order)(.) masked where the forward output was clamped  (bias_act.cu: G = 1, 2).
template <typename T>
__global__ __launch_bounds__(256) void bias_act_grad_kernel(const T* __restrict__ g, const T* __restrict__ b,
                                                            const T* __restrict__ xref, const T* __restrict__ yref,
                                                            const T* __restrict__ dy2, T* __restrict__ out, int64_t n,
                                                            int64_t size_b, int64_t step_b, int order, int act,
                                                            float alpha, float gain, float clamp) {
    using A = typename Acc<T>::type;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        A xr = xref ? (A)xref[i] : A(0);
        if (b) xr += (A)b[(i / step_b) % size_b];
        A yr = yref ? (A)yref[i] : A(0);
        if (act == 9) yr = (xr < A(-80) ? A(0) : xr / (exp(-xr) + A(1))) * (A)gain;      // swish keeps x, not y
        const A yy = gain != 0.f ? yr / (A)gain : A(0);
        A v = (A)g[i] * activate_deriv<A>(order, act, yy, xr, (A)alpha) * (A)gain;
        if (dy2) v *= (A)dy2[i];
        if (clamp >= 0.f && !(yr > -(A)clamp && yr < (A)clamp)) v = A(0);
        out[i] = (T)v;
    }
}

template <typename T>
int launch_grad(const void* g, const void* b, const void* xref, const void* yref, const void* dy2, void* out, int64_t n,
                int64_t size_b, int64_t step_b, int order, int act, float alpha, float gain, float clamp, hipStream_t st) {
    const int64_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? (want < 1 ? 1 : want) : 256 * 32);
    h3d::pre_launch();
    hipLaunchKernelGGL(bias_act_grad_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)g, (const T*)b, (const T*)xref,
                       (const T*)yref, (const T*)dy2, (T*)out, n, size_b, step_b, order, act, alpha, gain, clamp);
    return h3d::launch_status("h3d_bias_act_grad");
}

template <typename T>
int launch(const void* x, const void* b, void* y, int64_t n, int64_t size_b, int64_t step_b, int act, float alpha,
           float gain, float clamp, hipStream_t st) {
    const int64_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? (want < 1 ? 1 : want) : 256 * 32);
    h3d::pre_launch();
    hipLaunchKernelGGL(bias_act_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)x, (const T*)b, (T*)y, n, size_b, step_b,
                       act, alpha, gain, clamp);
    return h3d::launch_status("h3d_bias_act");
}

}  // namespace

extern "C" int h3d_bias_act(const void* x, const void* b, void* y, int64_t n, int dtype, int64_t size_b, int64_t step_b,
                            int act, float alpha, float gain, float clamp, h3d_stream_t stream) {
    H3D_REQUIRE(n >= 0, "h3d_bias_act: n < 0");
    if (n == 0) return H3D_OK;
    H3D_REQUIRE(x && y, "h3d_bias_act: null pointer");
    H3D_REQUIRE(act >= 1 && act <= 9, "h3d_bias_act: no kernel for activation index %d", act);
    H3D_REQUIRE(dtype >= 0 && dtype <= 2, "h3d_bias_act: dtype %d (0=f32,1=f16,2=f64)", dtype);
    H3D_REQUIRE(!b || (size_b >= 1 && step_b >= 1), "h3d_bias_act: bias given but size_b/step_b invalid");
    if (n == 0) return H3D_OK;
    if (!b) { size_b = 1; step_b = 1; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 0) {
        if ((n & 3) == 0 && h3d::aligned16(x) && h3d::aligned16(y) && (!b || (step_b & 3) == 0)) {
            const int64_t n4 = n / 4, want = (n4 + 255) / 256;
            const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);
            h3d::pre_launch();
            if (n * 4 >= (int64_t(64) << 20) && x != y)      // (in place the output lines are in the cache already: plain stores)
                hipLaunchKernelGGL(bias_act_f32x4<true>, dim3(grid), dim3(256), 0, st, (const float4*)x, (const float*)b, (float4*)y,
                                   n4, size_b, step_b, act, alpha, gain, clamp);
            else
                hipLaunchKernelGGL(bias_act_f32x4<false>, dim3(grid), dim3(256), 0, st, (const float4*)x, (const float*)b, (float4*)y,
                                   n4, size_b, step_b, act, alpha, gain, clamp);
            return h3d::launch_status("h3d_bias_act");
        }
        return launch<float>(x, b, y, n, size_b, step_b, act, alpha, gain, clamp, st);
    }
    if (dtype == 1) return launch<__half>(x, b, y, n, size_b, step_b, act, alpha, gain, clamp, st);
    return launch<double>(x, b, y, n, size_b, step_b, act, alpha, gain, clamp, st);
}

extern "C" int h3d_bias_act_grad(const void* g, const void* b, const void* xref, const void* yref, const void* dy2,
                                 void* out, int64_t n, int dtype, int64_t size_b, int64_t step_b, int order, int act,
                                 float alpha, float gain, float clamp, h3d_stream_t stream) {
    H3D_REQUIRE(n >= 0, "h3d_bias_act_grad: n < 0");
    if (n == 0) return H3D_OK;
    H3D_REQUIRE(g && out, "h3d_bias_act_grad: null pointer");
    H3D_REQUIRE(order == 1 || order == 2, "h3d_bias_act_grad: order must be 1 or 2 (got %d)", order);
    H3D_REQUIRE(act >= 1 && act <= 9, "h3d_bias_act_grad: no kernel for activation index %d", act);
    H3D_REQUIRE(dtype >= 0 && dtype <= 2, "h3d_bias_act_grad: dtype %d (0=f32,1=f16,2=f64)", dtype);
    H3D_REQUIRE(act == 9 ? xref != nullptr : ((act == 1 && clamp < 0.f) || yref != nullptr),
                "h3d_bias_act_grad: activation %d needs %s", act, act == 9 ? "xref" : "yref");
    H3D_REQUIRE(!b || (size_b >= 1 && step_b >= 1), "h3d_bias_act_grad: bias given but size_b/step_b invalid");
    H3D_REQUIRE(order == 1 || dy2, "h3d_bias_act_grad: order 2 needs dy");
    if (!b) { size_b = 1; step_b = 1; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 0) return launch_grad<float>(g, b, xref, yref, dy2, out, n, size_b, step_b, order, act, alpha, gain, clamp, st);
    if (dtype == 1) return launch_grad<__half>(g, b, xref, yref, dy2, out, n, size_b, step_b, order, act, alpha, gain, clamp, st);
    return launch_grad<double>(g, b, xref, yref, dy2, out, n, size_b, step_b, order, act, alpha, gain, clamp, st);
}
