// Hierarchical (coarse + fine) sampling support of the renderer, for gfx950.
// Reference: lib/generators/volume_rendering.py:261-303 (sample_pdf) and the re-sampling / merge block of
// Map3DGenerator.render, lib/generators/map3d_generator.py:449-509.  All three kernels are HBM / latency bound
// glue around the two field evaluations; the field itself and the integration are h3d_neural_field* / h3d_ray_integrate.
#include "common.hpp"

namespace {

constexpr int kMaxBins = 512;

// One wavefront per ray.  cdf is accumulated sequentially in fp32 like torch.cumsum; every lane then inverts the cdf for
// its samples with a binary search (torch.searchsorted, right=False) and the reference's linear interpolation.
__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights,
                                                         const float* __restrict__ u, float* __restrict__ out,
                                                         int64_t n_rays, int nb, int ns, float eps) {
    __shared__ float s_cdf[4][kMaxBins];
    __shared__ float s_bin[4][kMaxBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= n_rays) return;
    const int n = nb - 1;                                   // number of weights
    float* cdf = s_cdf[wave];
    float* bin = s_bin[wave];
    const float* w = weights + ray * n;
    float part = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float v = w[i] + eps;
        cdf[i + 1] = v;
        part += v;
    }
    for (int i = lane; i < nb; i += 64) bin[i] = bins[ray * nb + i];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float acc = 0.f;
        cdf[0] = 0.f;
        for (int i = 1; i <= n; ++i) {
            acc += cdf[i] / part;
            cdf[i] = acc;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < ns; j += 64) {
        const float uu = u[ray * ns + j];
        int lo = 0, hi = nb;                                // first index with cdf[idx] >= uu, nb if none
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] < uu) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 < 0 ? 0 : lo - 1;
        const int above = lo > n ? n : lo;
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = c1 - c0;
        if (denom < eps) denom = 1.f;
        const float b0 = bin[below], b1 = bin[above];
        out[ray * ns + j] = b0 + (uu - c0) / denom * (b1 - b0);
    }
}

// points[b][r*S + s] = origin[b] + dirs[b][r] * z[b][r][s]
__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ origin, const float* __restrict__ dirs,
                                                         const float* __restrict__ z, float* __restrict__ points,
                                                         int64_t total, int64_t RS, int S) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // sample index over [B, R, S]
    if (i >= total) return;
    const int64_t b = i / RS, ray = i / S;
    const float zz = z[i];
    const float* o = origin + b * 3;
    const float* d = dirs + ray * 3;
    points[i * 3 + 0] = fmaf(d[0], zz, o[0]);
    points[i * 3 + 1] = fmaf(d[1], zz, o[1]);
    points[i * 3 + 2] = fmaf(d[2], zz, o[2]);
}

// One workgroup per ray: stable rank of every depth among the ray's fine + coarse samples (fine first, as torch.cat),
// then the field rows are moved to their sorted positions.
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                    const float* __restrict__ fine_z, const float* __restrict__ coarse_z,
                                                    float* __restrict__ out, float* __restrict__ out_z, int Sf, int Sc, int C1) {
    __shared__ float zs[kMaxBins];
    __shared__ int src[kMaxBins];
    const int64_t ray = blockIdx.x;
    const int n = Sf + Sc, t = threadIdx.x;
    for (int i = t; i < n; i += 256) zs[i] = i < Sf ? fine_z[ray * Sf + i] : coarse_z[ray * Sc + (i - Sf)];
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        const float zi = zs[i];
        const bool ni = zi != zi;
        int rank = 0;
        // total order: by value, NaN after every number (as torch.sort), ties and NaNs by index -> ranks are always a
        // permutation, so every output row is written even for NaN depths
        for (int j = 0; j < n; ++j) {
            const float zj = zs[j];
            const bool nj = zj != zj;
            const bool before = ni ? (!nj || j < i) : (!nj && (zj < zi || (zj == zi && j < i)));
            rank += before ? 1 : 0;
        }
        src[rank] = i;
        out_z[ray * n + rank] = zi;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    if ((C1 & 3) == 0) {
        const int c4 = C1 >> 2;
        for (int r = wave; r < n; r += 4) {
            const int i = src[r];
            const float4* s4 = reinterpret_cast<const float4*>(i < Sf ? fine + (ray * Sf + i) * C1 : coarse + (ray * Sc + (i - Sf)) * C1);
            float4* d4 = reinterpret_cast<float4*>(out + (ray * n + r) * C1);
            for (int c = lane; c < c4; c += 64) d4[c] = s4[c];
        }
    } else {
        for (int r = wave; r < n; r += 4) {
            const int i = src[r];
            const float* s1 = i < Sf ? fine + (ray * Sf + i) * C1 : coarse + (ray * Sc + (i - Sf)) * C1;
            float* d1 = out + (ray * n + r) * C1;
            for (int c = lane; c < C1; c += 64) d1[c] = s1[c];
        }
    }
}

}  // namespace

extern "C" int h3d_sample_pdf(const float* bins, const float* weights, const float* u, float* samples, int64_t n_rays,
                              int n_bins, int n_samples, float eps, h3d_stream_t stream) {
    H3D_REQUIRE(n_rays >= 0 && n_samples >= 0, "h3d_sample_pdf: bad sizes");
    if (n_rays == 0 || n_samples == 0) return H3D_OK;            // empty tensors have no storage
    H3D_REQUIRE(bins && weights && u && samples, "h3d_sample_pdf: null pointer");
    H3D_REQUIRE(n_bins >= 2 && n_bins <= kMaxBins, "h3d_sample_pdf: n_bins=%d must be in [2, %d]", n_bins, kMaxBins);
    if (n_rays == 0 || n_samples == 0) return H3D_OK;
    const int64_t groups = (n_rays + 3) / 4;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_sample_pdf: too many rays");
    h3d::pre_launch();
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)groups), dim3(256), 0, static_cast<hipStream_t>(stream), bins, weights,
                       u, samples, n_rays, n_bins, n_samples, eps);
    return h3d::launch_status("h3d_sample_pdf");
}

extern "C" int h3d_ray_points(const float* origin, const float* dirs, const float* z_vals, float* points, int B, int64_t R,
                              int S, h3d_stream_t stream) {
    H3D_REQUIRE(B >= 0 && R >= 0 && S >= 1, "h3d_ray_points: bad sizes");
    const int64_t total = (int64_t)B * R * S;
    if (total == 0) return H3D_OK;
    H3D_REQUIRE(origin && dirs && z_vals && points, "h3d_ray_points: null pointer");
    const int64_t groups = (total + 255) / 256;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_ray_points: too many samples");
    h3d::pre_launch();
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)groups), dim3(256), 0, static_cast<hipStream_t>(stream), origin, dirs,
                       z_vals, points, total, R * S, S);
    return h3d::launch_status("h3d_ray_points");
}

extern "C" int h3d_merge_samples(const float* fine, const float* coarse, const float* fine_z, const float* coarse_z, float* out,
                                 float* out_z, int64_t n_rays, int Sf, int Sc, int C1, h3d_stream_t stream) {
    H3D_REQUIRE(n_rays >= 0 && Sf >= 0 && Sc >= 0 && C1 >= 1, "h3d_merge_samples: bad sizes");
    if (n_rays == 0 || Sf + Sc == 0) return H3D_OK;
    H3D_REQUIRE((Sf == 0 || (fine && fine_z)) && (Sc == 0 || (coarse && coarse_z)) && out && out_z, "h3d_merge_samples: null pointer");
    H3D_REQUIRE(Sf + Sc <= kMaxBins, "h3d_merge_samples: %d samples per ray exceed %d", Sf + Sc, kMaxBins);
    H3D_REQUIRE((C1 & 3) != 0 || (h3d::aligned16(fine) && h3d::aligned16(coarse) && h3d::aligned16(out)),
                "h3d_merge_samples: field tensors must be 16-byte aligned");
    H3D_REQUIRE(n_rays < (int64_t(1) << 31), "h3d_merge_samples: too many rays");
    h3d::pre_launch();
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)n_rays), dim3(256), 0, static_cast<hipStream_t>(stream), fine, coarse, fine_z,
                       coarse_z, out, out_z, Sf, Sc, C1);
    return h3d::launch_status("h3d_merge_samples");
}
