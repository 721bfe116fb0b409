// P3 modulated convolutions on the fp32 matrix cores (tile engine of field_common.hpp), for gfx950.
//
//   h3d_modconv1x1   per-pixel modulated 1x1 conv with demodulation
//                    == SpatialStyleModLayer.forward, lib/components/map3d_layers.py:60-80
//                        m = affine(style) + 1;  out = (x*m) W * rsqrt((m^2) W^2 + eps) + b
//                    three GEMMs per 64-pixel tile: style->m (K = S), (x*m) W and (m^2) W^2 (K = Cin).
//   h3d_modconv2d    StyleGAN2 modulated k x k conv with per-sample weights
//                    == StyleModLayer.forward_group_conv, lib/components/cips_layers.py:235-278
//                    evaluated in the algebraically identical "modulate the input, demodulate the output" form
//                        out[b,o] = d[b,o] * sum_{i,ky,kx} W[o,i,ky,kx] * (s[b,i] * x[b,i,.+ky,.+kx]) + bias[o]
//                    as an implicit GEMM: for every filter tap the shifted, modulated input slab [Cin][64 px] is
//                    staged in LDS and contracted with that tap's [Cin x Cout] weight slice; the per-sample vectors
//                    s = affine(style)+1 and d = rsqrt((W^2 summed over taps) s^2 + eps) are O(B*C) host-side GEMVs.
// Weights arrive packed by h3d_pack_matrix (MFMA B-fragment order).
#include "field_common.hpp"

using namespace h3d;

namespace {

struct Args1 {
    const float* x;        // [B*P, Cin]
    const float* style;    // [B*P, S]
    const float* w_aff;    // packed [S -> Cin]
    const float* b_aff;    // [CinP]  affine bias + 1
    const float* w;        // packed [Cin -> Cout]
    const float* w2;       // packed squares
    const float* bias;     // [CoutP]
    float* out;            // [B*P, Cout]
    int64_t rows;
    int Cin, Cout, S, CinP, CoutP, SP, demod;
    float eps;
};

template <int NTW>
__global__ __launch_bounds__(kFieldThreads) void modconv1x1_kernel(Args1 A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rowsA = A.CinP > A.SP ? A.CinP : A.SP;
    float* bufA = smem;                    // [max(SP, CinP)][MS]  style tile, then x*m
    float* bufB = bufA + rowsA * kMS;      // [CinP][MS]           m^2
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int NTi = A.CinP / 32, NTo = A.CoutP / 32;

    for (int idx = t; idx < 64 * A.SP; idx += kFieldThreads) {
        const int k = idx % A.SP, m = idx / A.SP;
        const int64_t r = r0 + m;
        bufA[k * kMS + m] = (k < A.S && r < A.rows) ? A.style[r * A.S + k] : 0.f;
    }
    __syncthreads();
    f32x16 acc[2][NTW];
    zero_acc<NTW>(acc);
    gemm_phase<NTW>(acc, bufA, reinterpret_cast<const float4*>(A.w_aff), A.SP / 8, 0, A.SP / 8, NTi, wave, lane);
    __syncthreads();
    // m = acc + (b + 1); write x*m -> bufA, m^2 -> bufB (x read straight from HBM in accumulator layout)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + 4 * i;
        if (nt >= NTi) continue;
        const int n = nt * 32 + j;
        const bool okn = n < A.Cin;
        const float b1 = A.b_aff[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float xm[4], mm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mt * 32 + rg * 8 + 4 * h + q;
                    const int64_t r = r0 + m;
                    const float mv = okn ? acc[mt][i][rg * 4 + q] + b1 : 0.f;
                    const float xv = (okn && r < A.rows) ? A.x[r * A.Cin + n] : 0.f;
                    xm[q] = xv * mv;
                    mm[q] = mv * mv;
                }
                *reinterpret_cast<float4*>(bufA + n * kMS + mt * 32 + rg * 8 + 4 * h) = make_float4(xm[0], xm[1], xm[2], xm[3]);
                *reinterpret_cast<float4*>(bufB + n * kMS + mt * 32 + rg * 8 + 4 * h) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            }
    }
    __syncthreads();
    f32x16 acc2[2][NTW];
    zero_acc<NTW>(acc);
    zero_acc<NTW>(acc2);
    gemm_phase<NTW>(acc, bufA, reinterpret_cast<const float4*>(A.w), A.CinP / 8, 0, A.CinP / 8, NTo, wave, lane);
    if (A.demod)
        gemm_phase<NTW>(acc2, bufB, reinterpret_cast<const float4*>(A.w2), A.CinP / 8, 0, A.CinP / 8, NTo, wave, lane);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + 4 * i;
        const int n = nt * 32 + j;
        if (nt >= NTo || n >= A.Cout) continue;
        const float bb = A.bias[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r >> 2) * 8 + 4 * h + (r & 3);
                const int64_t row = r0 + m;
                if (row >= A.rows) continue;
                float v = acc[mt][i][r];
                if (A.demod) v *= rsqrtf(acc2[mt][i][r] + A.eps);
                A.out[row * A.Cout + n] = v + bb;
            }
    }
}

struct Args2 {
    const float* x;        // [B, Cin, H, W]
    const float* smod;     // [B, CinP]  affine(style) + 1
    const float* dmod;     // [B, CoutP] demodulation (ones when disabled)
    const float* w;        // packed [k*k][Cin -> Cout]
    const float* bias;     // [CoutP]
    float* out;            // [B, Cout, H, W]
    int Cin, Cout, CinP, CoutP, H, W, k;
};

template <int NTW>
__global__ __launch_bounds__(kFieldThreads) void modconv2d_kernel(Args2 A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rowsA = A.CinP > A.CoutP ? A.CinP : A.CoutP;
    float* actT = smem;                     // [max(CinP, CoutP)][MS]: input slab per tap, then the output tile
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.y;
    const int HW = A.H * A.W, pad = A.k / 2;
    const int p0 = blockIdx.x * 64;
    const int NTo = A.CoutP / 32, KB = A.CinP / 8;
    const float* __restrict__ xb = A.x + (int64_t)b * A.Cin * HW;
    const float* __restrict__ sm = A.smod + (int64_t)b * A.CinP;
    (void)rowsA;

    f32x16 acc[2][NTW];
    zero_acc<NTW>(acc);
    for (int tap = 0; tap < A.k * A.k; ++tap) {
        const int ky = tap / A.k - pad, kx = tap % A.k - pad;
        __syncthreads();                    // previous tap's GEMM finished reading actT
        for (int idx = t; idx < 64 * A.CinP; idx += kFieldThreads) {
            const int m = idx & 63, i = idx >> 6;
            const int p = p0 + m;
            float v = 0.f;
            if (i < A.Cin && p < HW) {
                const int y = p / A.W + ky, xx = p % A.W + kx;
                if (y >= 0 && y < A.H && xx >= 0 && xx < A.W) v = xb[(int64_t)i * HW + y * A.W + xx] * sm[i];
            }
            actT[i * kMS + m] = v;
        }
        __syncthreads();
        gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(A.w) + (int64_t)tap * NTo * KB * 64, KB, 0, KB, NTo, wave, lane);
    }
    __syncthreads();
    {
        const float* __restrict__ dm = A.dmod + (int64_t)b * A.CoutP;
        const float* __restrict__ bs = A.bias;
        store_act<NTW>(acc, actT, NTo, A.Cout, wave, lane,
                       [&](int n) { return make_float2(dm[n], bs[n]); },
                       [](float v, const float2& c) { return fmaf(v, c.x, c.y); });
    }
    __syncthreads();
    // transposed, coalesced write-out: a row of the LDS tile is 64 consecutive pixels of one output channel
    float* __restrict__ ob = A.out + (int64_t)b * A.Cout * HW;
    for (int idx = t; idx < 64 * A.Cout; idx += kFieldThreads) {
        const int m = idx & 63, o = idx >> 6;
        if (p0 + m < HW) ob[(int64_t)o * HW + p0 + m] = actT[o * kMS + m];
    }
}

template <typename K, typename AT>
int launch_k(K kernel, const AT& A, dim3 grid, size_t lds, hipStream_t st, const char* what) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    h3d::pre_launch();
    hipLaunchKernelGGL(kernel, grid, dim3(kFieldThreads), lds, st, A);
    return h3d::launch_status(what);
}

}  // namespace

extern "C" int h3d_modconv1x1(const float* x, const float* style, const float* w_aff_packed, const float* b_aff_plus1,
                              const float* w_packed, const float* w2_packed, const float* bias, float* out, int64_t rows,
                              int Cin, int Cout, int S, int demodulate, float eps, h3d_stream_t stream) {
    H3D_REQUIRE(x && style && w_aff_packed && b_aff_plus1 && w_packed && bias && out, "h3d_modconv1x1: null pointer");
    H3D_REQUIRE(!demodulate || w2_packed, "h3d_modconv1x1: demodulation needs the squared weights");
    H3D_REQUIRE(rows >= 0 && Cin >= 1 && Cout >= 1 && S >= 1, "h3d_modconv1x1: bad shape");
    if (Cin > 256 || Cout > 256 || S > 256) {
        h3d::set_error("h3d_modconv1x1: widths up to 256 are built (got Cin=%d Cout=%d S=%d)", Cin, Cout, S);
        return H3D_EUNSUPPORTED;
    }
    if (rows == 0) return H3D_OK;
    Args1 A{};
    A.x = x; A.style = style; A.w_aff = w_aff_packed; A.b_aff = b_aff_plus1; A.w = w_packed; A.w2 = w2_packed; A.bias = bias;
    A.out = out; A.rows = rows; A.Cin = Cin; A.Cout = Cout; A.S = S; A.demod = demodulate; A.eps = eps;
    A.CinP = round_up(Cin, 32); A.CoutP = round_up(Cout, 32); A.SP = round_up(S, 32);
    const int rowsA = A.CinP > A.SP ? A.CinP : A.SP;
    const size_t lds = sizeof(float) * (size_t)(rowsA + A.CinP) * kMS;
    const int64_t tiles = (rows + 63) / 64;
    H3D_REQUIRE(tiles < (int64_t(1) << 31), "h3d_modconv1x1: too many rows");
    const int widest = A.CinP > A.CoutP ? A.CinP : A.CoutP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (widest <= 128) return launch_k(modconv1x1_kernel<1>, A, dim3((unsigned)tiles), lds, st, "h3d_modconv1x1");
    return launch_k(modconv1x1_kernel<2>, A, dim3((unsigned)tiles), lds, st, "h3d_modconv1x1");
}

extern "C" int h3d_modconv2d(const float* x, const float* smod, const float* dmod, const float* w_packed,
                             const float* bias, float* out, int B, int Cin, int Cout, int H, int W, int k,
                             h3d_stream_t stream) {
    H3D_REQUIRE(x && smod && dmod && w_packed && bias && out, "h3d_modconv2d: null pointer");
    H3D_REQUIRE(B >= 0 && B <= 65535 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "h3d_modconv2d: bad shape");
    H3D_REQUIRE(k >= 1 && (k & 1) == 1 && k <= 7, "h3d_modconv2d: odd kernel sizes 1..7 (got %d)", k);
    if (Cin > 512 || Cout > 512) {
        h3d::set_error("h3d_modconv2d: widths up to 512 are built (got Cin=%d Cout=%d)", Cin, Cout);
        return H3D_EUNSUPPORTED;
    }
    if (B == 0) return H3D_OK;
    Args2 A{};
    A.x = x; A.smod = smod; A.dmod = dmod; A.w = w_packed; A.bias = bias; A.out = out;
    A.Cin = Cin; A.Cout = Cout; A.H = H; A.W = W; A.k = k;
    A.CinP = round_up(Cin, 32); A.CoutP = round_up(Cout, 32);
    const int rowsA = A.CinP > A.CoutP ? A.CinP : A.CoutP;
    const size_t lds = sizeof(float) * (size_t)rowsA * kMS;
    const int64_t tiles = ((int64_t)H * W + 63) / 64;
    H3D_REQUIRE(tiles < (int64_t(1) << 31), "h3d_modconv2d: image too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)tiles, (unsigned)B);
    switch ((A.CoutP / 32 + 3) / 4) {
        case 1: return launch_k(modconv2d_kernel<1>, A, grid, lds, st, "h3d_modconv2d");
        case 2: return launch_k(modconv2d_kernel<2>, A, grid, lds, st, "h3d_modconv2d");
        case 3: return launch_k(modconv2d_kernel<3>, A, grid, lds, st, "h3d_modconv2d");
        default: return launch_k(modconv2d_kernel<4>, A, grid, lds, st, "h3d_modconv2d");
    }
}
