// A6 volume integration for gfx950 (HBM-bound streaming kernel).
//
// Reference semantics: lib/generators/volume_rendering.py:12-56 (ray_integration).
//
// One 256-thread workgroup per ray.
//   phase 1 (wave 0): densities of the S samples are gathered (one strided dword per sample), turned
//           into alpha, and the transmittance T_s = prod_{j<s}(1 - alpha_j + 1e-12) is a wavefront
//           product-scan (64 samples per step, carry across steps) -> w_s in LDS, depth by wave reduce.
//   phase 2 (all waves): the ray's S x (C+1) block is streamed once with 16-byte loads.  The block is
//           contiguous in HBM; thread t owns float4 column q = t % NQ of sample rows s = t / NQ + k*G, so
//           consecutive lanes read consecutive 16-byte slots across G whole rows per step (coalesced) and
//           every thread keeps one float4 accumulator.  G partial sums meet in LDS.
// Algorithmic bytes per ray: 4*(S*(C+1) [field] + S [z] + S [weights] + C [features] + 1 [depth]).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float density(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? x : log1pf(expf(x));   // F.softplus (beta=1, threshold=20)
    return fmaxf(x, 0.f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Wave 0 only.  Writes w[0..S) (with the last_back fix-up applied when requested), weights_out, depth.
// Returns the background term 1 - sum(w) computed BEFORE the last_back update (volume_rendering.py:36,48).
__device__ __forceinline__ float scan_weights(const float* __restrict__ sig_base, int64_t sig_stride,
                                              const float* __restrict__ z, const float* __restrict__ noise,
                                              float* w_lds, float* __restrict__ weights_out,
                                              float* __restrict__ depth_out, int S, int clamp_mode,
                                              int last_back, int lane) {
    float carry = 1.f, wsum = 0.f, dsum = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool ok = s < S;
        float f = 1.f, alpha = 0.f, zz = 0.f;
        if (ok) {
            float sg = sig_base[(int64_t)s * sig_stride];
            if (noise) sg += noise[s];
            zz = z[s];
            const float delta = (s == S - 1) ? 1e9f : z[s + 1] - zz;
            alpha = 1.f - expf(-delta * density(sg, clamp_mode));
            f = (1.f - alpha) + 1e-12f;
        }
        float incl = f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t = __shfl_up(incl, off, 64);
            if (lane >= off) incl *= t;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float w = alpha * (carry * excl);
        carry *= __shfl(incl, 63, 64);
        if (ok) w_lds[s] = w;
        wsum += w;
        dsum += w * zz;
    }
    wsum = wave_sum(wsum);
    dsum = wave_sum(dsum);
    const float bg = 1.f - wsum;
    const float z_last = z[S - 1];
    if (lane == 0) *depth_out = dsum + bg * z_last;          // both last_back variants agree on depth
    __builtin_amdgcn_wave_barrier();
    if (last_back && lane == 0) w_lds[S - 1] += bg;
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < S; s += 64) weights_out[s] = w_lds[s];
    return bg;
}

// float4 streaming variant: requires (C+1) % 4 == 0, NQ = (C+1)/4 <= 256, field 16-byte aligned.
__global__ __launch_bounds__(kThreads) void ray_integrate_vec4(
    const float* __restrict__ field, const float* __restrict__ z_vals, const float* __restrict__ noise,
    float* __restrict__ feats, float* __restrict__ depth, float* __restrict__ weights, int S, int C,
    int clamp_mode, int last_back, int white_back) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NQ = (C + 1) >> 2;
    const int G = kThreads / NQ;
    float* w_lds = smem;                                        // S floats (+pad to 4)
    float4* red = reinterpret_cast<float4*>(smem + ((S + 3) & ~3));   // G*NQ float4 (+1 float bg)
    float* bg_slot = reinterpret_cast<float*>(red + G * NQ);

    const int64_t ray = blockIdx.x;
    const int t = threadIdx.x;
    const int g = t / NQ, q = t - g * NQ;
    const bool active = g < G;
    const float4* __restrict__ blk = reinterpret_cast<const float4*>(field + ray * (int64_t)S * (C + 1));

    // issue the first loads before waiting for the weights
    constexpr int U = 4;
    float4 pre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int s = g + u * G;
        pre[u] = (active && s < S) ? blk[(int64_t)s * NQ + q] : make_float4(0, 0, 0, 0);
    }

    if (t < 64) {
        const float bg = scan_weights(field + ray * (int64_t)S * (C + 1) + C, C + 1, z_vals + ray * S,
                                      noise ? noise + ray * S : nullptr, w_lds, weights + ray * S, depth + ray, S,
                                      clamp_mode, last_back, t);
        if (t == 0) *bg_slot = bg;
    }
    __syncthreads();

    float4 acc = make_float4(0, 0, 0, 0);
    if (active) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = g + u * G;
            if (s < S) {
                const float w = w_lds[s];
                acc.x += w * pre[u].x; acc.y += w * pre[u].y; acc.z += w * pre[u].z; acc.w += w * pre[u].w;
            }
        }
#pragma unroll 4
        for (int s = g + U * G; s < S; s += G) {
            const float4 v = blk[(int64_t)s * NQ + q];
            const float w = w_lds[s];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        red[g * NQ + q] = acc;
    }
    __syncthreads();
    if (t < NQ) {
        float4 r = red[t];
        for (int k = 1; k < G; ++k) {
            const float4 o = red[k * NQ + t];
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        const float add = white_back ? *bg_slot : 0.f;
        float* o = feats + ray * (int64_t)C;
        const int c = t * 4;
        if (c + 0 < C) o[c + 0] = r.x + add;
        if (c + 1 < C) o[c + 1] = r.y + add;
        if (c + 2 < C) o[c + 2] = r.z + add;
        if (c + 3 < C) o[c + 3] = r.w + add;
    }
}

// Generic fallback (any C, any alignment): thread per channel, dword loads.
__global__ __launch_bounds__(kThreads) void ray_integrate_scalar(
    const float* __restrict__ field, const float* __restrict__ z_vals, const float* __restrict__ noise,
    float* __restrict__ feats, float* __restrict__ depth, float* __restrict__ weights, int S, int C,
    int clamp_mode, int last_back, int white_back) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w_lds = smem;
    float* bg_slot = smem + S;
    const int64_t ray = blockIdx.x;
    const int t = threadIdx.x;
    const float* __restrict__ blk = field + ray * (int64_t)S * (C + 1);
    if (t < 64) {
        const float bg = scan_weights(blk + C, C + 1, z_vals + ray * S, noise ? noise + ray * S : nullptr, w_lds,
                                      weights + ray * S, depth + ray, S, clamp_mode, last_back, t);
        if (t == 0) *bg_slot = bg;
    }
    __syncthreads();
    const float add = white_back ? *bg_slot : 0.f;
    for (int c = t; c < C; c += kThreads) {
        float acc = 0.f;
#pragma unroll 4
        for (int s = 0; s < S; ++s) acc += w_lds[s] * blk[(int64_t)s * (C + 1) + c];
        feats[ray * (int64_t)C + c] = acc + add;
    }
}

}  // namespace

extern "C" int h3d_ray_integrate(const float* field, const float* z_vals, const float* noise, float* feats,
                                 float* depth, float* weights, int64_t n_rays, int S, int C, int clamp_mode,
                                 int last_back, int white_back, h3d_stream_t stream) {
    H3D_REQUIRE(field && z_vals && feats && depth && weights, "h3d_ray_integrate: null pointer");
    H3D_REQUIRE(n_rays >= 0 && n_rays < (int64_t(1) << 31), "h3d_ray_integrate: n_rays=%lld out of range", (long long)n_rays);
    H3D_REQUIRE(S >= 1 && S <= 8192, "h3d_ray_integrate: S=%d must be in [1,8192]", S);
    H3D_REQUIRE(C >= 1, "h3d_ray_integrate: C=%d must be >= 1", C);
    H3D_REQUIRE(clamp_mode == 0 || clamp_mode == 1, "h3d_ray_integrate: clamp_mode must be 0 (relu) or 1 (softplus)");
    if (n_rays == 0) return H3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int NQ = (C + 1) / 4;
    const bool vec = ((C + 1) % 4 == 0) && NQ <= kThreads && h3d::aligned16(field);
    if (vec) {
        const int G = kThreads / NQ;
        const size_t lds = sizeof(float) * (((S + 3) & ~3) + 4 * G * NQ + 4);
        h3d::pre_launch();
        hipLaunchKernelGGL(ray_integrate_vec4, dim3((unsigned)n_rays), dim3(kThreads), lds, st, field, z_vals, noise,
                           feats, depth, weights, S, C, clamp_mode, last_back, white_back);
    } else {
        const size_t lds = sizeof(float) * (S + 4);
        h3d::pre_launch();
        hipLaunchKernelGGL(ray_integrate_scalar, dim3((unsigned)n_rays), dim3(kThreads), lds, st, field, z_vals,
                           noise, feats, depth, weights, S, C, clamp_mode, last_back, white_back);
    }
    return h3d::launch_status("h3d_ray_integrate");
}
