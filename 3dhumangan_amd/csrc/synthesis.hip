// A7 + A8 + A9: the SPADE synthesis network as ONE kernel per 64-pixel tile, for gfx950.
//
// Reference semantics (eval mode): lib/generators/map3d_generator.py:58-97 (SynthesisNetwork.forward),
// lib/components/map3d_layers.py:176-190 (SPADE2d), :218-238 (SPADEBlock), :260-275 (SynthesisInput),
// :346-352 (ToRGB); bilinear F.interpolate at map3d_generator.py:244-245.
//
// What is algebraically folded on the host (exactly, no approximation -- see lib/generators/synthesis_pack.py):
//   * spectral norm: conv weight = weight_orig / sigma(u, v) with the stored u, v;
//   * eval BatchNorm: per-channel scale/shift;
//   * SPADE with a spatially constant style (every block outside mod_blocks): gamma/beta are per-sample
//     vectors, so BN + modulation collapse into a per-(sample, channel) affine "ab" applied on the A-operand
//     read path of the following conv GEMM (together with the leaky ReLU);
//   * SPADE with a per-pixel style (mod_blocks): the shared 1x1 conv is linear and commutes with bilinear
//     interpolation (weights sum to one), so it runs once at render resolution (a library GEMM producing the
//     low-resolution map G); here its 128 channels are bilinearly sampled per output pixel, biased by the
//     per-sample constant (conv bias + fixed-style term), ReLU'd, and pushed through the gamma / beta GEMMs.
// What runs here per tile, all on the fp32 matrix cores with activations resident in LDS / registers:
//   x0 = sin(W_in * (i, j) + b)                                                     (A8, VALU, in registers)
//   per block: [SPADE_0 -> lrelu -> conv_0 -> SPADE_1 -> lrelu -> conv_1 (+ skip)] , ToRGB accumulated
// Only the 3-channel image is written to HBM.  MFMA-bound: 2*(18*C^2 + 6*256*C) flop per pixel for the shipped
// 9-block / 3-mod-block configs (the 128-wide shared convs left the per-pixel path).
#include "field_common.hpp"

using namespace h3d;

namespace {

constexpr int kShared = 128;   // hidden width of SPADE's shared MLP (map3d_layers.py:169)

struct Args {
    const float* blob;
    h3d_synth_desc D;
    const float* G;      // [B, Hr*Wr, g_channels] low-res shared-conv maps (channels last)
    const float* cst;    // [B, n_cst, 128]
    const float* ab;     // [B, n_ab, 2, HdP]
    float* rgb;          // [B, 3, H, W]
    int g_channels, Hr, Wr, n_cst, n_ab, H, W, HdP, C;
};

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, 0.2f * v); }

__device__ __forceinline__ float linspace_pm1(int n, int i) {   // torch.linspace(-1, 1, n)[i]
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

template <int NTW>
__global__ __launch_bounds__(kFieldThreads) void synthesis_kernel(Args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HdP = A.HdP, C = A.C;
    const int NT = HdP / 32, KBH = HdP / 8;
    float* actT = smem;                         // [HdP][MS]   raw activations / conv inputs
    float* aT = actT + HdP * kMS;               // [128][MS]   ReLU'd shared-MLP activations of the current SPADE
    float* abT = aT + kShared * kMS;            // [2][HdP]    per-sample affine of a constant-style SPADE
    float* part = aT;                           // [4][3][64]  ToRGB partial sums: aliases aT (only live between the
                                                //             barriers around a block's ToRGB, when no SPADE is running)
    float* ci = abT + 2 * HdP;                  // [64] pixel coordinate i (rows), then j
    float* cj = ci + 64;
    int* tap = reinterpret_cast<int*>(cj + 64); // [64][4] low-res tap offsets (pixel index) ; weights follow
    float* tw = reinterpret_cast<float*>(tap + 256);   // [64][2] (ty, tx)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int64_t HW = (int64_t)A.H * A.W;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const float* __restrict__ blob = A.blob;
    const h3d_synth_desc& D = A.D;

    // ---- per-pixel geometry: synthesis-input coordinates and bilinear taps into the low-res maps
    if (t < 64) {
        int64_t p = p0 + t;
        if (p >= HW) p = HW - 1;
        const int Y = (int)(p / A.W), X = (int)(p % A.W);
        ci[t] = linspace_pm1(A.H, Y);
        cj[t] = linspace_pm1(A.W, X);
        float sy = ((float)Y + 0.5f) * ((float)A.Hr / (float)A.H) - 0.5f;
        float sx = ((float)X + 0.5f) * ((float)A.Wr / (float)A.W) - 0.5f;
        sy = fmaxf(sy, 0.f);
        sx = fmaxf(sx, 0.f);
        const int y0 = min((int)sy, A.Hr - 1), x0 = min((int)sx, A.Wr - 1);
        const int y1 = min(y0 + 1, A.Hr - 1), x1 = min(x0 + 1, A.Wr - 1);
        tap[t * 4 + 0] = y0 * A.Wr + x0;
        tap[t * 4 + 1] = y0 * A.Wr + x1;
        tap[t * 4 + 2] = y1 * A.Wr + x0;
        tap[t * 4 + 3] = y1 * A.Wr + x1;
        tw[t * 2 + 0] = sy - (float)y0;
        tw[t * 2 + 1] = sx - (float)x0;
    }
    __syncthreads();

    // ---- A8: x0[n][m] = sin(w[n][0]*i + w[n][1]*j + b[n]) straight into the accumulator layout
    f32x16 xr[2][NTW], xres[2][NTW];
    {
        const float* __restrict__ win = blob + D.w_in;
        const float* __restrict__ bin = blob + D.b_in;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int nt = wave + 4 * i;
            const int n = min(nt, NT - 1) * 32 + j;
            const bool ok = nt < NT && n < C;
            const float w0 = ok ? win[n] : 0.f, w1 = ok ? win[HdP + n] : 0.f, bb = ok ? bin[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float4 vi = *reinterpret_cast<const float4*>(ci + mt * 32 + rg * 8 + 4 * h);
                    const float4 vj = *reinterpret_cast<const float4*>(cj + mt * 32 + rg * 8 + 4 * h);
                    xr[mt][i][rg * 4 + 0] = ok ? sin_accurate(w0 * vi.x + w1 * vj.x + bb) : 0.f;
                    xr[mt][i][rg * 4 + 1] = ok ? sin_accurate(w0 * vi.y + w1 * vj.y + bb) : 0.f;
                    xr[mt][i][rg * 4 + 2] = ok ? sin_accurate(w0 * vi.z + w1 * vj.z + bb) : 0.f;
                    xr[mt][i][rg * 4 + 3] = ok ? sin_accurate(w0 * vi.w + w1 * vj.w + bb) : 0.f;
                }
        }
    }
    zero_acc<NTW>(xres);
    float rgb_acc = 0.f;                 // threads < 192: (channel t>>6, pixel t&63)
    bool have_rgb = false;
    bool raw_in_lds = false;             // does actT currently hold the raw block input?

    for (int blk = 0; blk < D.n_blocks; ++blk) {
        const h3d_block_desc& Bk = D.block[blk];
        if (Bk.skip) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < NTW; ++i) xres[mt][i] = xr[mt][i];
        }
        for (int s = 0; s < 2; ++s) {
            const h3d_spade_desc& Sp = Bk.spade[s];
            f32x16 acc[2][NTW];
            zero_acc<NTW>(acc);
            if (Sp.pixel_style) {
                // ---- shared-MLP activations: bilinear sample of G + per-sample constant, ReLU -> aT[k][m]
                {
                    const int m = lane, k0 = wave * 32;
                    const float* __restrict__ Gb = A.G + (int64_t)b * A.Hr * A.Wr * A.g_channels + Sp.g_offset + k0;
                    const float* __restrict__ cs = A.cst + ((int64_t)b * A.n_cst + Sp.cst_index) * kShared + k0;
                    const float ty = tw[m * 2], tx = tw[m * 2 + 1];
                    const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
                    const float4* g00 = reinterpret_cast<const float4*>(Gb + (int64_t)tap[m * 4 + 0] * A.g_channels);
                    const float4* g01 = reinterpret_cast<const float4*>(Gb + (int64_t)tap[m * 4 + 1] * A.g_channels);
                    const float4* g10 = reinterpret_cast<const float4*>(Gb + (int64_t)tap[m * 4 + 2] * A.g_channels);
                    const float4* g11 = reinterpret_cast<const float4*>(Gb + (int64_t)tap[m * 4 + 3] * A.g_channels);
#pragma unroll 2
                    for (int q = 0; q < 8; ++q) {
                        const float4 a = g00[q], bq = g01[q], c = g10[q], d = g11[q];
                        const float4 k4 = *reinterpret_cast<const float4*>(cs + q * 4);
                        // same association as F.interpolate: lerp in x on both rows, then lerp in y
                        const float tx1 = 1.f - tx, ty1 = 1.f - ty;
                        (void)w00; (void)w01; (void)w10; (void)w11;
                        float v0 = (a.x * tx1 + bq.x * tx) * ty1 + (c.x * tx1 + d.x * tx) * ty + k4.x;
                        float v1 = (a.y * tx1 + bq.y * tx) * ty1 + (c.y * tx1 + d.y * tx) * ty + k4.y;
                        float v2 = (a.z * tx1 + bq.z * tx) * ty1 + (c.z * tx1 + d.z * tx) * ty + k4.z;
                        float v3 = (a.w * tx1 + bq.w * tx) * ty1 + (c.w * tx1 + d.w * tx) * ty + k4.w;
                        aT[(k0 + q * 4 + 0) * kMS + m] = fmaxf(v0, 0.f);
                        aT[(k0 + q * 4 + 1) * kMS + m] = fmaxf(v1, 0.f);
                        aT[(k0 + q * 4 + 2) * kMS + m] = fmaxf(v2, 0.f);
                        aT[(k0 + q * 4 + 3) * kMS + m] = fmaxf(v3, 0.f);
                    }
                }
                __syncthreads();
                // ---- gamma / beta GEMMs (K = 128) and the modulation, element-wise in accumulator layout
                f32x16 accB[2][NTW];
                zero_acc<NTW>(accB);
                gemm_phase<NTW>(acc, aT, reinterpret_cast<const float4*>(blob + Sp.w_gamma), kShared / 8, 0, kShared / 8, NT, wave, lane);
                gemm_phase<NTW>(accB, aT, reinterpret_cast<const float4*>(blob + Sp.w_beta), kShared / 8, 0, kShared / 8, NT, wave, lane);
                const float* __restrict__ vec = blob + Sp.vec;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int nt = wave + 4 * i;
                    const int n = min(nt, NT - 1) * 32 + j;
                    const float g1 = vec[n], bt = vec[HdP + n], sc = vec[2 * HdP + n], sh = vec[3 * HdP + n];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float nrm = fmaf(xr[mt][i][r], sc, sh);
                            acc[mt][i][r] = lrelu(fmaf(nrm, acc[mt][i][r] + g1, accB[mt][i][r] + bt));
                        }
                }
                // actT may still be read by other waves (previous conv) -- the barrier above covers it because
                // every wave passed its previous GEMM before building aT.
                store_act<NTW>(acc, actT, NT, C, wave, lane, [](int) { return 0; }, [](float v, int) { return v; });
                __syncthreads();
                zero_acc<NTW>(acc);
                gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + Sp.w_conv), KBH, 0, KBH, NT, wave, lane);
            } else {
                // ---- constant-style SPADE: raw activations are in actT, the affine + lrelu rides on the A path
                if (!raw_in_lds) {
                    __syncthreads();
                    store_act<NTW>(xr, actT, NT, C, wave, lane, [](int) { return 0; }, [](float v, int) { return v; });
                }
                const float* __restrict__ abg = A.ab + ((int64_t)b * A.n_ab + Sp.ab_index) * 2 * HdP;
                for (int idx = t; idx < 2 * HdP; idx += kFieldThreads) abT[idx] = abg[idx];
                __syncthreads();
                gemm_phase<NTW, true>(acc, actT, reinterpret_cast<const float4*>(blob + Sp.w_conv), KBH, 0, KBH, NT, wave, lane, abT, HdP);
            }
            // ---- conv epilogue: bias (+ skip), result becomes the next raw activation
            {
                const float* __restrict__ bc = blob + Sp.b_conv;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int nt = wave + 4 * i;
                    const int n = min(nt, NT - 1) * 32 + j;
                    const float bias = bc[n];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[mt][i][r] + bias;
                            if (s == 1 && Bk.skip) v += xres[mt][i][r];
                            xr[mt][i][r] = v;
                        }
                }
            }
            raw_in_lds = false;
            // does anything need the raw value in LDS next?  (constant-style SPADE or ToRGB)
            const bool next_const = (s == 0) ? !Bk.spade[1].pixel_style
                                             : (blk + 1 < D.n_blocks && !D.block[blk + 1].spade[0].pixel_style);
            if (next_const || (s == 1 && Bk.to_rgb)) {
                __syncthreads();         // every wave finished reading actT in the conv GEMM
                store_act<NTW>(xr, actT, NT, C, wave, lane, [](int) { return 0; }, [](float v, int) { return v; });
                raw_in_lds = true;
                if (s == 1 && Bk.to_rgb) __syncthreads();
            }
        }
        if (Bk.to_rgb) {
            const float* __restrict__ wr = blob + Bk.w_rgb;
            const int kq = HdP / 4, k0 = wave * kq;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < kq; ++k) {
                const float x = actT[(k0 + k) * kMS + lane];
                s0 = fmaf(x, wr[k0 + k], s0);
                s1 = fmaf(x, wr[HdP + k0 + k], s1);
                s2 = fmaf(x, wr[2 * HdP + k0 + k], s2);
            }
            part[wave * 192 + lane] = s0;
            part[wave * 192 + 64 + lane] = s1;
            part[wave * 192 + 128 + lane] = s2;
            __syncthreads();
            if (t < 192) {
                const int c = t >> 6;
                const float v = ((part[t] + part[192 + t]) + (part[384 + t] + part[576 + t])) + wr[3 * HdP + c];
                rgb_acc = have_rgb ? v + rgb_acc : v;
            }
            have_rgb = true;
            __syncthreads();             // part is reused by the next ToRGB
        }
    }
    if (t < 192) {
        const int c = t >> 6, m = t & 63;
        const int64_t p = p0 + m;
        if (p < HW) A.rgb[((int64_t)b * 3 + c) * HW + p] = rgb_acc;
    }
}

size_t lds_bytes(int HdP) {
    return sizeof(float) * ((size_t)HdP * kMS + kShared * kMS + 2 * HdP + 128 + 256 + 128);
}

template <int NTW>
int launch_one(const Args& A, int B, int64_t tiles, hipStream_t st) {
    H3D_ALLOW_MAX_LDS((synthesis_kernel<NTW>));
    h3d::pre_launch();
    hipLaunchKernelGGL((synthesis_kernel<NTW>), dim3((unsigned)tiles, (unsigned)B), dim3(kFieldThreads), lds_bytes(A.HdP), st, A);
    return h3d::launch_status("h3d_synthesis");
}

}  // namespace

extern "C" int h3d_pack_matrix(const float* w, int ld_in, int in_begin, int in_count, int n_out, int KB, int NT,
                               float* dst) {
    H3D_REQUIRE(w && dst, "h3d_pack_matrix: null pointer");
    H3D_REQUIRE(ld_in >= 1 && in_begin >= 0 && in_count >= 0 && n_out >= 0 && KB >= 1 && NT >= 1, "h3d_pack_matrix: bad shape");
    pack_matrix(w, ld_in, in_begin, in_count, n_out, KB, NT, dst);
    return H3D_OK;
}

extern "C" int h3d_synthesis(const void* blob, const h3d_synth_desc* desc, const float* G, int g_channels, int Hr,
                             int Wr, const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H,
                             int W, h3d_stream_t stream) {
    H3D_REQUIRE(blob && desc && rgb, "h3d_synthesis: null pointer");
    H3D_REQUIRE(h3d::aligned16(blob), "h3d_synthesis: blob must be 16-byte aligned");
    H3D_REQUIRE(desc->n_blocks >= 1 && desc->n_blocks <= H3D_MAX_BLOCKS, "h3d_synthesis: n_blocks=%d", desc->n_blocks);
    H3D_REQUIRE(desc->C >= 1, "h3d_synthesis: C=%d", desc->C);
    H3D_REQUIRE(B >= 0 && B <= 65535 && H >= 1 && W >= 1, "h3d_synthesis: bad output shape");
    bool any_pixel = false, any_const = false;
    for (int k = 0; k < desc->n_blocks; ++k)
        for (int s = 0; s < 2; ++s) {
            const h3d_spade_desc& sp = desc->block[k].spade[s];
            if (sp.pixel_style) {
                any_pixel = true;
                H3D_REQUIRE(sp.g_offset >= 0 && sp.g_offset + kShared <= g_channels && (sp.g_offset & 3) == 0,
                            "h3d_synthesis: block %d spade %d g_offset out of range", k, s);
                H3D_REQUIRE(sp.cst_index >= 0 && sp.cst_index < n_cst, "h3d_synthesis: cst_index out of range");
            } else {
                any_const = true;
                H3D_REQUIRE(sp.ab_index >= 0 && sp.ab_index < n_ab, "h3d_synthesis: ab_index out of range");
            }
        }
    H3D_REQUIRE(!any_pixel || (G && cst && Hr >= 1 && Wr >= 1 && (g_channels & 3) == 0 && h3d::aligned16(G) && h3d::aligned16(cst)),
                "h3d_synthesis: per-pixel style blocks need G/cst (16-byte aligned, channels %% 4 == 0)");
    H3D_REQUIRE(!any_const || ab, "h3d_synthesis: constant-style blocks need the ab table");
    H3D_REQUIRE(desc->block[desc->n_blocks - 1].to_rgb, "h3d_synthesis: the last block must feed ToRGB");
    if (B == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const float*>(blob);
    A.D = *desc;
    A.G = G; A.cst = cst; A.ab = ab; A.rgb = rgb;
    A.g_channels = g_channels; A.Hr = Hr; A.Wr = Wr; A.n_cst = n_cst; A.n_ab = n_ab; A.H = H; A.W = W;
    A.C = desc->C;
    A.HdP = round_up(desc->C, 32);
    const int64_t tiles = ((int64_t)H * W + 63) / 64;
    H3D_REQUIRE(tiles < (int64_t(1) << 31), "h3d_synthesis: image too large");
    H3D_REQUIRE(lds_bytes(A.HdP) <= 160 * 1024, "h3d_synthesis: width %d does not fit the 160 KB LDS", desc->C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch ((A.HdP / 32 + 3) / 4) {
        case 1: return launch_one<1>(A, B, tiles, st);
        case 2: return launch_one<2>(A, B, tiles, st);
        case 3: return launch_one<3>(A, B, tiles, st);
        case 4: return launch_one<4>(A, B, tiles, st);
        default:
            h3d::set_error("h3d_synthesis: width %d exceeds the 512 this build supports", desc->C);
            return H3D_EUNSUPPORTED;
    }
}
