// A7 bilinear resize (align_corners=False) for gfx950, NCHW fp32.
// Reference call site: lib/generators/map3d_generator.py:244-245, 324-325 (F.interpolate(..., mode='bilinear')).
// HBM-bound: each output element is written once; the 4 source taps come from L1/L2 (the source plane is
// 16-28x smaller than the destination).  One thread produces 4 consecutive output pixels of a row (16-byte store).
// bilinear_rows (wide images): a thread walks 8 output rows of its 4-pixel column group and keeps the two horizontally
// interpolated source rows in registers -- when upsampling, consecutive output rows mostly share them, so the gathers per
// 16-byte store drop from 16 to ~4 -- with a 3-D grid (no 64-bit index division).  Same arithmetic, same association, same bits
// as bilinear_kernel.
#include "common.hpp"

namespace {

__device__ __forceinline__ void src_index(int dst, float ratio, int n_in, int& i0, int& i1, float& t) {
    float s = ((float)dst + 0.5f) * ratio - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = min((int)s, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    t = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w,
                                                       int H, int W, float ry, float rx, int64_t planes) {
    const int Wq = (W + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = planes * H * Wq;
    if (i >= total) return;
    const int xq = (int)(i % Wq);
    const int Y = (int)((i / Wq) % H);
    const int64_t pl = i / ((int64_t)Wq * H);
    int y0, y1;
    float ty;
    src_index(Y, ry, h, y0, y1, ty);
    const float* __restrict__ r0 = in + (pl * h + y0) * (int64_t)w;
    const float* __restrict__ r1 = in + (pl * h + y1) * (int64_t)w;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = min(xq * 4 + k, W - 1);
        int x0, x1;
        float tx;
        src_index(X, rx, w, x0, x1, tx);
        const float top = r0[x0] * (1.f - tx) + r0[x1] * tx;
        const float bot = r1[x0] * (1.f - tx) + r1[x1] * tx;
        v[k] = top * (1.f - ty) + bot * ty;
    }
    float* o = out + (pl * H + Y) * (int64_t)W + xq * 4;
    if ((W & 3) == 0 && h3d::aligned16(out)) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (xq * 4 + k < W) o[k] = v[k];
    }
}

constexpr int kRows = 8;            // output rows per thread of bilinear_rows

__global__ __launch_bounds__(256) void bilinear_rows(const float* __restrict__ in, float* __restrict__ out, int h, int w, int H, int W,
                                                     float ry, float rx, int vec_ok) {
    const int xq = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kRows;
    if (xq * 4 >= W || Y0 >= H) return;
    const float* __restrict__ src = in + (int64_t)blockIdx.z * h * w;
    float* __restrict__ dst = out + (int64_t)blockIdx.z * H * W;
    int x0[4], x1[4];
    float tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) src_index(min(xq * 4 + k, W - 1), rx, w, x0[k], x1[k], tx[k]);
    auto hrow = [&](int yi, float (&v)[4]) {
        const float* __restrict__ r = src + (int64_t)yi * w;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = r[x0[k]] * (1.f - tx[k]) + r[x1[k]] * tx[k];
    };
    float top[4], bot[4];
    int ytop = -1, ybot = -1;
#pragma unroll 1
    for (int j = 0; j < kRows; ++j) {
        const int Y = Y0 + j;
        if (Y >= H) break;
        int y0, y1;
        float ty;
        src_index(Y, ry, h, y0, y1, ty);
        if (y0 != ytop) {
            if (y0 == ybot) {
#pragma unroll
                for (int k = 0; k < 4; ++k) top[k] = bot[k];
            } else {
                hrow(y0, top);
            }
            ytop = y0;
        }
        if (y1 != ybot) {
            if (y1 == ytop) {
#pragma unroll
                for (int k = 0; k < 4; ++k) bot[k] = top[k];
            } else {
                hrow(y1, bot);
            }
            ybot = y1;
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = top[k] * (1.f - ty) + bot[k] * ty;
        float* o = dst + (int64_t)Y * W + xq * 4;
        if (vec_ok) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (xq * 4 + k < W) o[k] = v[k];
        }
    }
}

}  // namespace

extern "C" int h3d_bilinear_resize(const float* in, float* out, int B, int C, int h, int w, int H, int W,
                                   h3d_stream_t stream) {
    H3D_REQUIRE(in && out, "h3d_bilinear_resize: null pointer");
    H3D_REQUIRE(B >= 0 && C >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "h3d_bilinear_resize: bad shape");
    if (B == 0) return H3D_OK;
    const int64_t planes = (int64_t)B * C;
    if (W >= 128 && H >= 2 * kRows && (int64_t)H * W < (int64_t(1) << 31) && (int64_t)h * w < (int64_t(1) << 31)) {
        const int vec_ok = (W % 4 == 0) && h3d::aligned16(out);
        const int Wq = (W + 3) / 4;
        for (int64_t z0 = 0; z0 < planes; z0 += 65535) {
            const unsigned nz = (unsigned)((planes - z0) < 65535 ? (planes - z0) : 65535);
            h3d::pre_launch();
            hipLaunchKernelGGL(bilinear_rows, dim3((Wq + 63) / 64, (H + 4 * kRows - 1) / (4 * kRows), nz), dim3(256), 0,
                               static_cast<hipStream_t>(stream), in + z0 * h * w, out + z0 * H * W, h, w, H, W, (float)h / (float)H,
                               (float)w / (float)W, vec_ok);
            const int rc = h3d::launch_status("h3d_bilinear_resize");
            if (rc) return rc;
        }
        return H3D_OK;
    }
    const int64_t total = planes * H * ((W + 3) / 4);
    const int64_t grid = (total + 255) / 256;
    H3D_REQUIRE(grid < (int64_t(1) << 31), "h3d_bilinear_resize: tensor too large");
    h3d::pre_launch();
    hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, h, w,
                       H, W, (float)h / (float)H, (float)w / (float)W, planes);
    return h3d::launch_status("h3d_bilinear_resize");
}
