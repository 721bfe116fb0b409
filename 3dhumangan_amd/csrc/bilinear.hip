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

template <bool NT>
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
            // Streaming (non-temporal) stores for an output that does not fit the caches anyway (round 6): the resized map is
            // written once and never read back here; with plain stores the write stream ran at 3.6 TB/s, with `nt` at 6.1 TB/s
            // on the same lease (profiles/r6_ops_nt_stores.txt).  NT (a template parameter: a run-time branch between the two store
            // kinds is merged into one plain store by the optimiser) is chosen for outputs >= 64 MB.
            typedef float f4 __attribute__((ext_vector_type(4)));
            if constexpr (NT) __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(o));
            else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (xq * 4 + k < W) o[k] = v[k];
        }
    }
}

// ---- channels-last maps [B, h, w, C] (the training path's layout; C % 4 == 0) ------------------------------------------------
// forward: a thread writes 4 channels of one output pixel (16-byte store; the four taps are 16-byte loads from the small source)
// RELU: the ReLU behind the resize in the same pass (the SPADEs' shared map, lib/components/map3d_layers.py:170-174: mlp_shared = conv + ReLU)
template <bool RELU>
__global__ __launch_bounds__(256) void bilinear_cl_fwd(const float* __restrict__ in, float* __restrict__ out, int h, int w, int H, int W,
                                                       int C4, float ry, float rx) {
    const int c4 = blockIdx.x * 64 + (threadIdx.x & 63);
    const int X = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c4 >= C4 || X >= W) return;
    const int Y = blockIdx.z % H, b = blockIdx.z / H;
    int y0, y1, x0, x1;
    float ty, tx;
    src_index(Y, ry, h, y0, y1, ty);
    src_index(X, rx, w, x0, x1, tx);
    const float4* __restrict__ src = reinterpret_cast<const float4*>(in) + (int64_t)b * h * w * C4 + c4;
    const float4 a = src[((int64_t)y0 * w + x0) * C4], bq = src[((int64_t)y0 * w + x1) * C4];
    const float4 c = src[((int64_t)y1 * w + x0) * C4], d = src[((int64_t)y1 * w + x1) * C4];
    auto mix = [&](float a_, float b_, float c_, float d_) {
        const float top = a_ * (1.f - tx) + b_ * tx, bot = c_ * (1.f - tx) + d_ * tx;       // same association as bilinear_kernel
        const float v = top * (1.f - ty) + bot * ty;
        return RELU ? (v > 0.f ? v : 0.f) : v;
    };
    reinterpret_cast<float4*>(out)[(((int64_t)b * H + Y) * W + X) * C4 + c4] =
        make_float4(mix(a.x, bq.x, c.x, d.x), mix(a.y, bq.y, c.y, d.y), mix(a.z, bq.z, c.z, d.z), mix(a.w, bq.w, c.w, d.w));
}

// adjoint along ONE axis: dst[o, i, r] = sum_I weight(I -> i) src[o, I, r] for src [O, N, R4] float4 rows, dst [O, n, R4].
// A thread owns (o, r) and walks I = 0 .. N-1 once: the source index i0(I) is non-decreasing, so two running sums (rows i0 and
// i0 + 1) are enough and each finished row is written exactly once -- the source is read once, coalesced over r, no atomics.
// MASK: src is the gradient of relu(resize(x)) and `mask` (same shape) the forward's output: rows are read as src * (mask > 0)
template <bool MASK>
__global__ __launch_bounds__(256) void bilinear_cl_adjoint_axis(const float* __restrict__ src, const float* __restrict__ mask,
                                                                float* __restrict__ dst, int N, int n, int64_t R4, int64_t total, float ratio) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t o = idx / R4, r = idx - o * R4;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(src) + o * N * R4 + r;
    const float4* __restrict__ m = MASK ? reinterpret_cast<const float4*>(mask) + o * N * R4 + r : nullptr;
    float4* __restrict__ d = reinterpret_cast<float4*>(dst) + o * n * R4 + r;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    int cur = 0;
    auto row = [&](int I) {
        float4 q = s[(int64_t)I * R4];
        if constexpr (MASK) {
            const float4 k = m[(int64_t)I * R4];
            q.x = k.x > 0.f ? q.x : 0.f; q.y = k.y > 0.f ? q.y : 0.f; q.z = k.z > 0.f ? q.z : 0.f; q.w = k.w > 0.f ? q.w : 0.f;
        }
        return q;
    };
    float4 v = row(0);
    for (int I = 0; I < N; ++I) {
        const float4 nxt = I + 1 < N ? row(I + 1) : v;                                      // one row ahead of the arithmetic
        int i0, i1;
        float t;
        src_index(I, ratio, n, i0, i1, t);
        while (cur < i0) {
            d[(int64_t)cur * R4] = acc0;
            acc0 = acc1;
            acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
            ++cur;
        }
        const float w0 = 1.f - t;
        acc0.x += w0 * v.x; acc0.y += w0 * v.y; acc0.z += w0 * v.z; acc0.w += w0 * v.w;
        if (i1 == i0) { acc0.x += t * v.x; acc0.y += t * v.y; acc0.z += t * v.z; acc0.w += t * v.w; }
        else          { acc1.x += t * v.x; acc1.y += t * v.y; acc1.z += t * v.z; acc1.w += t * v.w; }
        v = nxt;
    }
    for (; cur < n; ++cur) {
        d[(int64_t)cur * R4] = acc0;
        acc0 = acc1;
        acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace

// Channels-last variant: in [B,h,w,C] -> out [B,H,W,C] (C a multiple of 4, 16-byte aligned), same arithmetic as h3d_bilinear_resize.
static int resize_cl_any(bool relu, const float* in, float* out, int B, int h, int w, int H, int W, int C, h3d_stream_t stream) {
    H3D_REQUIRE(in && out, "h3d_bilinear_resize_cl: null pointer");
    H3D_REQUIRE(B >= 0 && C >= 4 && C % 4 == 0 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "h3d_bilinear_resize_cl: bad shape (C must be a multiple of 4)");
    H3D_REQUIRE(h3d::aligned16(in) && h3d::aligned16(out), "h3d_bilinear_resize_cl: operands must be 16-byte aligned");
    H3D_REQUIRE((int64_t)B * H < 65536, "h3d_bilinear_resize_cl: B * H must be below 65536");
    if (B == 0) return H3D_OK;
    const int C4 = C / 4;
    h3d::pre_launch();
    const dim3 grid((C4 + 63) / 64, (W + 3) / 4, B * H);
    if (relu)
        hipLaunchKernelGGL(bilinear_cl_fwd<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, out, h, w, H, W, C4,
                           (float)h / (float)H, (float)w / (float)W);
    else
        hipLaunchKernelGGL(bilinear_cl_fwd<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, out, h, w, H, W, C4,
                           (float)h / (float)H, (float)w / (float)W);
    return h3d::launch_status("h3d_bilinear_resize_cl");
}
static int resize_cl_bwd_any(const float* dout, const float* mask, float* tmp, float* din, int B, int h, int w, int H, int W, int C,
                             h3d_stream_t stream) {
    H3D_REQUIRE(dout && tmp && din, "h3d_bilinear_resize_cl_bwd: null pointer");
    H3D_REQUIRE(B >= 0 && C >= 4 && C % 4 == 0 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "h3d_bilinear_resize_cl_bwd: bad shape (C must be a multiple of 4)");
    H3D_REQUIRE(h3d::aligned16(dout) && h3d::aligned16(tmp) && h3d::aligned16(din) && (!mask || h3d::aligned16(mask)),
                "h3d_bilinear_resize_cl_bwd: operands must be 16-byte aligned");
    if (B == 0) return H3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t C4 = C / 4;
    {   // rows: [B][H][W*C4] -> [B][h][W*C4]
        const int64_t R4 = (int64_t)W * C4, total = (int64_t)B * R4;
        H3D_REQUIRE((total + 255) / 256 < (int64_t(1) << 31), "h3d_bilinear_resize_cl_bwd: tensor too large");
        h3d::pre_launch();
        if (mask)
            hipLaunchKernelGGL(bilinear_cl_adjoint_axis<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dout, mask, tmp, H, h, R4,
                               total, (float)h / (float)H);
        else
            hipLaunchKernelGGL(bilinear_cl_adjoint_axis<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dout, nullptr, tmp, H, h,
                               R4, total, (float)h / (float)H);
        const int rc = h3d::launch_status("h3d_bilinear_resize_cl_bwd");
        if (rc) return rc;
    }
    {   // columns: [B*h][W][C4] -> [B*h][w][C4]
        const int64_t R4 = C4, total = (int64_t)B * h * R4;
        h3d::pre_launch();
        hipLaunchKernelGGL(bilinear_cl_adjoint_axis<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, nullptr, din, W, w, R4,
                           total, (float)w / (float)W);
        return h3d::launch_status("h3d_bilinear_resize_cl_bwd");
    }
}
extern "C" int h3d_bilinear_resize_cl(const float* in, float* out, int B, int h, int w, int H, int W, int C, h3d_stream_t stream) {
    return resize_cl_any(false, in, out, B, h, w, H, W, C, stream);
}
// Adjoint of h3d_bilinear_resize_cl: dout [B,H,W,C] -> din [B,h,w,C]; tmp: B*h*W*C floats of scratch (rows first, then columns).
extern "C" int h3d_bilinear_resize_cl_bwd(const float* dout, float* tmp, float* din, int B, int h, int w, int H, int W, int C,
                                          h3d_stream_t stream) {
    return resize_cl_bwd_any(dout, nullptr, tmp, din, B, h, w, H, W, C, stream);
}
// relu(resize(in)) in one pass, and its gradient: din = resize^T(dout * (out > 0)) with `out` the forward's result (round 6: the
// ReLU of the SPADEs' shared map and its mask no longer make passes of their own over the [B, H, W, 128 n] tensor).
extern "C" int h3d_bilinear_resize_cl_relu(const float* in, float* out, int B, int h, int w, int H, int W, int C, h3d_stream_t stream) {
    return resize_cl_any(true, in, out, B, h, w, H, W, C, stream);
}
extern "C" int h3d_bilinear_resize_cl_relu_bwd(const float* dout, const float* out, float* tmp, float* din, int B, int h, int w, int H, int W,
                                               int C, h3d_stream_t stream) {
    H3D_REQUIRE(out, "h3d_bilinear_resize_cl_relu_bwd: null pointer");
    return resize_cl_bwd_any(dout, out, tmp, din, B, h, w, H, W, C, stream);
}

extern "C" int h3d_bilinear_resize(const float* in, float* out, int B, int C, int h, int w, int H, int W,
                                   h3d_stream_t stream) {
    H3D_REQUIRE(in && out, "h3d_bilinear_resize: null pointer");
    H3D_REQUIRE(B >= 0 && C >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "h3d_bilinear_resize: bad shape");
    if (B == 0) return H3D_OK;
    const int64_t planes = (int64_t)B * C;
    if (W >= 128 && H >= 2 * kRows && (int64_t)H * W < (int64_t(1) << 31) && (int64_t)h * w < (int64_t(1) << 31)) {
        const int vec_ok = (W % 4 == 0) && h3d::aligned16(out);
        const bool streaming = vec_ok && planes * H * W * 4 >= (int64_t(64) << 20);      // outputs beyond the caches: streaming stores (bilinear_rows)
        const int Wq = (W + 3) / 4;
        for (int64_t z0 = 0; z0 < planes; z0 += 65535) {
            const unsigned nz = (unsigned)((planes - z0) < 65535 ? (planes - z0) : 65535);
            h3d::pre_launch();
            const dim3 grid((Wq + 63) / 64, (H + 4 * kRows - 1) / (4 * kRows), nz);
            if (streaming) hipLaunchKernelGGL(bilinear_rows<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in + z0 * h * w,
                                           out + z0 * H * W, h, w, H, W, (float)h / (float)H, (float)w / (float)W, vec_ok);
            else hipLaunchKernelGGL(bilinear_rows<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in + z0 * h * w,
                                    out + z0 * H * W, h, w, H, W, (float)h / (float)H, (float)w / (float)W, vec_ok);
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
