// A7 + A8 + A9 on the bf16 matrix cores with split ("x3") operands and LDS-resident activations ("x3t"), for widths
// the register-resident engine (synthesis_x3.hip, <= 256) cannot hold: MAP3DBN (384), MAP3DBN512L (420), any C <= 448.
//
// Reference semantics (eval mode): lib/generators/map3d_generator.py:58-97 (SynthesisNetwork.forward),
// lib/components/map3d_layers.py:176-190 (SPADE2d), :218-238 (SPADEBlock), :260-275 (SynthesisInput), :346-352 (ToRGB);
// bilinear F.interpolate at map3d_generator.py:244-245.  Same exact host-side folding and the same descriptor as
// synthesis.hip (spectral norm, eval BatchNorm, constant-style SPADE -> per-(sample, channel) affine, shared 1x1 conv of
// the per-pixel SPADEs moved to render resolution); engine: x3t_common.hpp.
//
// One workgroup = 64 pixels (two 32-pixel sample tiles).  A wave owns a quarter of the channels of both tiles: the raw
// activations x stay in its accumulator registers (lane = pixel, registers = channels) from the coordinate input to the
// last ToRGB; every SPADE writes lrelu(modulated x) as bf16 hi / lo B fragments into LDS (lane-local 16-byte writes,
// accumulator-order K), every conv / gamma / beta is a GEMM over those fragments with its weights streamed from L2 in
// A-fragment order.  Only the 3-channel image reaches HBM.
#include "x3t_common.hpp"
#include <type_traits>

using namespace h3d;

namespace {

constexpr int kShared = 128;          // hidden width of SPADE's shared MLP (map3d_layers.py:169)
constexpr int kKSA = kShared / 16;    // k-steps of the gamma / beta GEMMs

struct Args {
    const unsigned char* wblob;   // bf16 hi/lo A fragments, descriptor w_gamma / w_beta / w_conv = BYTE offsets
    const float* tables;          // fp32, descriptor vec / b_conv / w_rgb / w_in / b_in = FLOAT offsets, vectors HdP long
    h3d_synth_desc D;
    const float* G;               // [B, Hr*Wr, g_channels] low-res shared-conv maps (channels last)
    const float* cst;             // [B, n_cst, 128]
    const float* ab;              // [B, n_ab, 2, HdP]
    float* rgb;                   // [B, 3, H, W]
    int g_channels, Hr, Wr, n_cst, n_ab, H, W, NT, first_skip;
    int* ovf;                     // x2 tier: int[B], ovf[b] set to 1 when an activation of sample b leaves the range of the f16 planes (nullable)
    const int* run_if;            // int[B] (nullable): sample b is skipped when run_if[b] == 0 -- the guarded fallback of the x2 tier
};

__device__ __forceinline__ float lrelu(float v) { return vmax(v, 0.2f * v); }

__device__ __forceinline__ float linspace_pm1(int n, int i) {   // torch.linspace(-1, 1, n)[i]
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// Per-block view of the kernel state.  Rebuilt at the top of every block iteration from LAUNDERED copies of the weight /
// table pointers, the wave's tile ids and the lane id: none of the per-phase fragment addresses is then loop-invariant
// for the compiler, which would otherwise hoist the first k-steps' weight loads of every GEMM (and their addresses) out
// of the block loop and spill hundreds of registers (LICM).
// T / P: operand type and partial products per operand pair (x3t_common.hpp): BF16 / 3 is the fp32-class default; F16 / 2
// (weights hi + lo, activations one f16 value) and F16 / 1 (plain f16 matrix-core arithmetic) are the reduced-precision
// tiers of BASELINE config 5 -- f16 because one bf16 value (8 significant bits) per activation is too coarse.
// F16 / 4 is the x2 arithmetic (x3_common.hpp: one f16 product + one block-scaled fp6 product per contraction; fragments' "lo"
// planes hold the K-tiles' fp6 records, each pixel's record with its own power-of-two scale).
template <int NTF, int NX, typename T, int P>
struct Block {
    static constexpr int NU = 2 * NTF + NX;
    static constexpr bool LO = P == 3;          // activations carry a lo half
    static constexpr bool X2 = P == 4;          // activations carry fp6 records
    typedef typename std::conditional<std::is_same<T, F16>::value, SplitF16, SplitBF16>::type Split;
    const Args& A;
    X3tUnits<NTF, NX> U;
    const unsigned char* wblob;
    const float* tables;
    unsigned char *actT, *aT;
    float *part, *tw;
    int* tap;
    int lane, m, h, wave, b, t, KS, HdP, act_stride;
    float* gmax;                        // x2: this lane's running maximum of |activation| (range guard)
    Split split;
    X3tRing<NTF + NX>& ring;            // weight fragments in flight (x3t_common.hpp)

    // Visit the wave's tiles; fn(slot constant, tile, first unit, number of units (2 sample tiles, or 1 for the extra
    // unit)).  The two units of a full tile share every per-channel table value: tables are fetched once per TILE.
    template <typename FN>
    __device__ __forceinline__ void for_tiles(FN fn) const {
        static_for<0, NTF + NX>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            fn(ic, U.nt[i], i < NTF ? 2 * i : 2 * NTF, i < NTF ? 2 : 1);
            __builtin_amdgcn_sched_barrier(0);      // bound the hoisting of table loads to one tile
        });
    }
    __device__ __forceinline__ int unit_mt(int u) const { return U.mt(u); }
    template <typename F>
    __device__ __forceinline__ void store_unit(const f32x16& v, int nt, int mt, F f) const {
        if constexpr (X2) x3t_store_unit_x2<true>(v, actT, KS, nt, mt, lane, f, gmax);
        else x3t_store_unit<LO>(v, actT, KS, nt, mt, lane, split, f);
    }

    // constant-style SPADE of `src`: y = lrelu(x * a + b) (per-(sample, channel) affine from the host) -> actT
    __device__ __forceinline__ void store_const(f32x16 (&src)[NU], const h3d_spade_desc& Sp) const {
        const float* __restrict__ abg = A.ab + ((int64_t)b * A.n_ab + Sp.ab_index) * 2 * HdP + 4 * h;
        f32x4 sa[2][4], sb[2][4];                  // the tile's affine, fetched one tile ahead (global memory: L2 latency)
        auto fetch = [&](int slot, int nt) __attribute__((always_inline)) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                sa[slot][rg] = ld4(abg + nt * 32 + rg * 8);
                sb[slot][rg] = ld4(abg + HdP + nt * 32 + rg * 8);
            }
        };
        fetch(0, U.nt[0]);
        for_tiles([&](auto ic, int nt, int u0, int nu) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, sl = i & 1;
            if constexpr (i + 1 < NTF + NX) fetch(sl ^ 1, U.nt[i + 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nu) {
                    const int u = u0 + k;
                    pin1(src[u]);       // accumulator sets live in AGPRs; VALU code reads / writes them one unit at a time
                    store_unit(src[u], nt, unit_mt(u), [&](int rg, f32x4 v) {
                        f32x4 y;
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = lrelu(fmaf(v[q], sa[sl][rg][q], sb[sl][rg][q]));
                        return y;
                    });
                }
            }
        });
    }
    // dst = vec (ADD = false) or dst += vec (ADD = true), vec a per-channel vector of the tables
    template <bool ADD>
    __device__ __forceinline__ void add_vec(f32x16 (&dst)[NU], const float* __restrict__ vecp) const {
        for_tiles([&](auto, int nt, int u0, int nu) __attribute__((always_inline)) {
            f32x4 bb[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) bb[rg] = ld4(vecp + nt * 32 + rg * 8 + 4 * h);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nu) {
                    const int u = u0 + k;
                    if (ADD) pin1(dst[u]);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[u][rg * 4 + q] = ADD ? dst[u][rg * 4 + q] + bb[rg][q] : bb[rg][q];
                    pin1(dst[u]);
                }
            }
        });
    }
    __device__ __forceinline__ const unsigned char* conv_w(const h3d_spade_desc& Sp) const {
#ifdef H3D_EXPERIMENT_ALIAS_W        // timing experiment (wrong results): every convolution reads the first one's bytes (L2-resident)
        return wblob + A.D.block[0].spade[0].w_conv;
#else
        return wblob + Sp.w_conv;
#endif
    }
    // Request the first k-steps of this SPADE's convolution weights into the (idle) register ring: called BEFORE the epilogue and
    // the barrier in front of the convolution, so that their L2 round trip (~2 000 cycles, once per GEMM: 30 GEMMs per tile) hides
    // behind that work instead of stalling the GEMM's first MFMA (round 6; the field kernel has done this since round 3).
    __device__ __forceinline__ void prefetch_conv(const h3d_spade_desc& Sp) const {
#ifndef H3D_X3T_NO_PREFETCH
        x3t_prefetch<NTF, NX, P>(ring, conv_w(Sp), x3t_tile_bytes<X2>(KS), 0, U, lane);
#endif
    }
    // dst (+)= bias + Wconv * actT   (the ring holds the requests of prefetch_conv)
    template <bool ADD>
    __device__ __forceinline__ void conv(f32x16 (&dst)[NU], const h3d_spade_desc& Sp) const {
        add_vec<ADD>(dst, tables + Sp.b_conv);
#ifndef H3D_X3T_NO_PREFETCH
        gemm_x3t<T, NTF, NX, false, false, true, P>(dst, actT, act_stride, conv_w(Sp), x3t_tile_bytes<X2>(KS), 0, KS, U, lane, ring);
#else
        gemm_x3t<T, NTF, NX, false, false, false, P>(dst, actT, act_stride, conv_w(Sp), x3t_tile_bytes<X2>(KS), 0, KS, U, lane, ring);
#endif
    }
    // per-pixel-style SPADE: fragments of lrelu((x*sc + sh) * (1 + gamma) + beta) -> actT; g is the gamma / beta scratch
    __device__ __forceinline__ void store_pixel(f32x16 (&x)[NU], f32x16 (&g)[NU], const h3d_spade_desc& Sp) const {
#ifndef H3D_X3T_NO_PREFETCH
        constexpr bool kPre = true;
        x3t_prefetch<NTF, NX, P>(ring, wblob + Sp.w_gamma, x3t_tile_bytes<X2>(kKSA), 0, U, lane);      // gamma's first k-steps: under the bilinear gathers below
#else
        constexpr bool kPre = false;
#endif
        // ---- shared-MLP activations a = relu(bilinear(G) + cst) as B fragments (natural K order): wave w covers
        //      channels 32w .. 32w+31 = k-steps 2w, 2w+1; lane = pixel
        {
            const int px = lane, mt = px >> 5, pm = px & 31;
            const float* __restrict__ Gb = A.G + (int64_t)b * A.Hr * A.Wr * A.g_channels + Sp.g_offset;
            const float* __restrict__ cs = A.cst + ((int64_t)b * A.n_cst + Sp.cst_index) * kShared;
            const float ty = tw[px * 2], tx = tw[px * 2 + 1];
            const float tx1 = 1.f - tx, ty1 = 1.f - ty;
            const float* g00 = Gb + (int64_t)tap[px * 4 + 0] * A.g_channels;
            const float* g01 = Gb + (int64_t)tap[px * 4 + 1] * A.g_channels;
            const float* g10 = Gb + (int64_t)tap[px * 4 + 2] * A.g_channels;
            const float* g11 = Gb + (int64_t)tap[px * 4 + 3] * A.g_channels;
            u32x4 qh[4], ql[4];                            // x2: the K-tile's four fragments of this pixel (q = 2 j + hh)
            float amax[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {                  // (k-step, half) = (2w + q/2, q%2): 8 channels each
                const int ks = 2 * wave + (q >> 1), hh = q & 1, k0 = 16 * ks + 8 * hh;
                float v[8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 a = ld4(g00 + k0 + 4 * j), bq = ld4(g01 + k0 + 4 * j), c = ld4(g10 + k0 + 4 * j),
                                d = ld4(g11 + k0 + 4 * j), k4 = ld4(cs + k0 + 4 * j);
                    // same association as F.interpolate: lerp in x on both rows, then lerp in y
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[4 * j + i] = vrelu((a[i] * tx1 + bq[i] * tx) * ty1 + (c[i] * tx1 + d[i] * tx) * ty + k4[i]);
                }
                u32x4 hi, lo;
                if constexpr (X2) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        unsigned l2;
                        hi[e / 2] = split2_x2(v[e], v[e + 1], l2);
                        lo[e / 2] = l2;
                        amax[hh] = vmax3_abs2(amax[hh], v[e], v[e + 1]);
                    }
                    qh[q] = hi; ql[q] = lo;
                    *reinterpret_cast<u32x4*>(aT + x3t_frag(kKSA, mt, ks, 0) + (32 * hh + pm) * 16) = hi;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { unsigned l2; hi[e / 2] = split(v[e], v[e + 1], l2); lo[e / 2] = l2; }
                    unsigned char* dst = aT + x3t_frag(kKSA, mt, ks, 0) + (32 * hh + pm) * 16;
                    *reinterpret_cast<u32x4*>(dst) = hi;
                    *reinterpret_cast<u32x4*>(dst + 1024) = lo;
                }
                __builtin_amdgcn_sched_barrier(0);         // 10 x 16-byte loads in flight per step are plenty
            }
            if constexpr (X2) {
                // the records of fragment lanes (pm, hh = 0) and (pm, hh = 1) of K-tile `wave`: this pixel's k-steps 2w, 2w+1
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    i32x8 rec = x2_record_dyn(__builtin_bit_cast(F16::vec8, ql[hh]), __builtin_bit_cast(F16::vec8, ql[2 + hh]),
                                              __builtin_bit_cast(F16::vec8, qh[hh]), __builtin_bit_cast(F16::vec8, qh[2 + hh]), amax[hh]);
                    *gmax = vmax(*gmax, amax[hh]);
                    rec[7] = 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        *reinterpret_cast<u32x4*>(aT + x3t_frag(kKSA, mt, 2 * wave + j, 1) + (32 * hh + pm) * 16) =
                            u32x4{(unsigned)rec[4 * j], (unsigned)rec[4 * j + 1], (unsigned)rec[4 * j + 2], (unsigned)rec[4 * j + 3]};
                }
            }
        }
        H3D_TRACE(30);
        __syncthreads();
        H3D_TRACE(31);
        // ---- gamma: g = (1 + bias_gamma) + Wg a ;  g <- (x*sc + sh) * g + bias_beta ;  beta: g += Wb a
        const float* __restrict__ vec = tables + Sp.vec;
        constexpr int a_stride = kKSA * 2048;
        add_vec<false>(g, vec);
        gemm_x3t<T, NTF, NX, false, false, kPre, P>(g, aT, a_stride, wblob + Sp.w_gamma, x3t_tile_bytes<X2>(kKSA), 0, kKSA, U, lane, ring);
        if constexpr (kPre) x3t_prefetch<NTF, NX, P>(ring, wblob + Sp.w_beta, x3t_tile_bytes<X2>(kKSA), 0, U, lane);       // beta's: under the affine pass below
        H3D_TRACE(32);
        for_tiles([&](auto, int nt, int u0, int nu) __attribute__((always_inline)) {
            f32x4 bt[4], sc[4], sh[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8 + 4 * h;
                bt[rg] = ld4(vec + HdP + n); sc[rg] = ld4(vec + 2 * HdP + n); sh[rg] = ld4(vec + 3 * HdP + n);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nu) {
                    const int u = u0 + k;
                    pin1(x[u]); pin1(g[u]);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            g[u][rg * 4 + q] = fmaf(fmaf(x[u][rg * 4 + q], sc[rg][q], sh[rg][q]), g[u][rg * 4 + q], bt[rg][q]);
                    pin1(g[u]);
                }
            }
        });
        H3D_TRACE(33);
        gemm_x3t<T, NTF, NX, false, false, kPre, P>(g, aT, a_stride, wblob + Sp.w_beta, x3t_tile_bytes<X2>(kKSA), 0, kKSA, U, lane, ring);
        prefetch_conv(Sp);                                                          // the convolution's: under the fragment stores below
        H3D_TRACE(34);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            pin1(g[u]);
            store_unit(g[u], U.tile(u), U.mt(u), [&](int, f32x4 v) {
                f32x4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = lrelu(v[i]);
                return y;
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ToRGB of this block: rgb_acc (threads < 192: channel t>>6, pixel t&63) += Wrgb x + b, an fp32 dot product over this
    // lane's channels of both of its pixels (sample tiles), summed over the lane halves and the four waves
    __device__ __forceinline__ void to_rgb(f32x16 (&x)[NU], const h3d_block_desc& Bk, float& rgb_acc) const {
        const float* __restrict__ wr = tables + Bk.w_rgb;
        float pr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        for_tiles([&](auto, int nt, int u0, int nu) __attribute__((always_inline)) {
            f32x4 w0[4], w1[4], w2[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8 + 4 * h;
                w0[rg] = ld4(wr + n); w1[rg] = ld4(wr + HdP + n); w2[rg] = ld4(wr + 2 * HdP + n);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nu) {
                    const int u = u0 + k;
                    pin1(x[u]);
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v = x[u][rg * 4 + q];
                            s0 = fmaf(v, w0[rg][q], s0);
                            s1 = fmaf(v, w1[rg][q], s1);
                            s2 = fmaf(v, w2[rg][q], s2);
                        }
                    if (u < 2 * NTF) {
                        pr[k][0] += s0; pr[k][1] += s1; pr[k][2] += s2;          // unit 2i + k covers sample tile k
                    } else {                          // the extra unit covers sample tile xmt only
                        const float f0 = U.xmt == 0 ? 1.f : 0.f, f1 = 1.f - f0;
                        pr[0][0] = fmaf(s0, f0, pr[0][0]); pr[0][1] = fmaf(s1, f0, pr[0][1]); pr[0][2] = fmaf(s2, f0, pr[0][2]);
                        pr[1][0] = fmaf(s0, f1, pr[1][0]); pr[1][1] = fmaf(s1, f1, pr[1][1]); pr[1][2] = fmaf(s2, f1, pr[1][2]);
                    }
                }
            }
        });
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = pr[mt][c] + __shfl_xor(pr[mt][c], 32, 64);
                if (h == 0) part[wave * 192 + c * 64 + mt * 32 + m] = v;
            }
        __syncthreads();
        if (t < 192) rgb_acc += ((part[t] + part[192 + t]) + (part[384 + t] + part[576 + t])) + wr[3 * HdP + (t >> 6)];
        __syncthreads();                      // part is reused by the next ToRGB
    }
};

template <int NTF, int NX, typename T, int P>
__global__ __launch_bounds__(256, 1) void synthesis_x3t_kernel(Args A) {
    constexpr int NU = 2 * NTF + NX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NT = A.NT, KS = 2 * NT, HdP = 32 * NT;
    const int act_stride = KS * 2048;
    unsigned char* actT = smem_raw;                                   // [2][KS][2][1 KB]   conv inputs
    unsigned char* aT = actT + 2 * act_stride;                        // [2][8][2][1 KB]    shared-MLP activations
    float* part = reinterpret_cast<float*>(aT + 2 * kKSA * 2048);     // [4 waves][3][64]   ToRGB partial sums
    float* ci = part + 4 * 3 * 64;                                    // [64] pixel coordinate i, then j
    float* cj = ci + 64;
    int* tap = reinterpret_cast<int*>(cj + 64);                       // [64][4] low-res tap offsets (pixel index)
    float* tw = reinterpret_cast<float*>(tap + 256);                  // [64][2] (ty, tx)

    if (A.run_if && A.run_if[blockIdx.y] == 0) return;          // guarded fallback: nothing to redo for this sample
    H3D_TRACE_INIT();
    H3D_TRACE(0);
    const int t = threadIdx.x, lane0 = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane0 & 31, h = lane0 >> 5;
    const int b = blockIdx.y;
    const int64_t HW = (int64_t)A.H * A.W;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    float gmax = 0.f;                                // x2 tier: largest |activation| this lane converted (range guard)
    const h3d_synth_desc& D = A.D;
    X3tUnits<NTF, NX> U0;
    U0.init(wave);

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

    H3D_TRACE(1);
    f32x16 cur[NU];                     // raw activations of this wave's units (lane = pixel, registers = channels)
    X3tRing<NTF + NX> ring;
    float rgb_acc = 0.f;                // threads < 192: (channel t>>6, pixel t&63)

    // ---- A8: x0[n][p] = sin(w0[n]*i + w1[n]*j + b[n]) straight into the accumulator layout
    {
        const float* __restrict__ win = A.tables + D.w_in;
        const float* __restrict__ bin = A.tables + D.b_in;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int nt = U0.tile(u), px = U0.mt(u) * 32 + m;
            const float vi = ci[px], vj = cj[px];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8 + 4 * h;
                const f32x4 w0 = ld4(win + n), w1 = ld4(win + HdP + n), bb = ld4(bin + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[u][rg * 4 + i] = sin_hw(w0[i] * vi + w1[i] * vj + bb[i]);
            }
            pin1(cur[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    H3D_TRACE(2);
    auto block_view = [&]() __attribute__((always_inline)) {
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        X3tUnits<NTF, NX> U = U0;
#pragma unroll
        for (int i = 0; i < NTF + NX; ++i) asm volatile("" : "+s"(U.nt[i]));
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        return Block<NTF, NX, T, P>{A, U, A.wblob + opaque, A.tables + opaque, actT, aT, part, tw, tap,
                              lane, m, h, wave, b, t, KS, HdP, act_stride, &gmax, {}, ring};
    };

    // ================= blocks before the first skip connection (either style) =======================================
    // conv <- SPADE(cur), twice; cur is replaced by each conv's output.  `acc` only lives inside a per-pixel SPADE.
#pragma unroll 1
    for (int blk = 0; blk < A.first_skip; ++blk) {
        const h3d_block_desc& Bk = D.block[blk];
        const Block<NTF, NX, T, P> K = block_view();
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            const h3d_spade_desc& Sp = Bk.spade[s];
            H3D_TRACE(10);
            if (Sp.pixel_style) {
                f32x16 acc[NU];
                K.store_pixel(cur, acc, Sp);          // (requests the convolution's first weights itself, behind its own GEMMs)
            } else {
                K.prefetch_conv(Sp);
                K.store_const(cur, Sp);
            }
            H3D_TRACE(11);
            __syncthreads();
            H3D_TRACE(12);
            K.template conv<false>(cur, Sp);          // the old cur is dead: it went into the fragments
            H3D_TRACE(13);
            __syncthreads();                  // every wave finished reading actT / aT before the next stage rewrites them
            H3D_TRACE(14);
        }
        if (Bk.to_rgb) K.to_rgb(cur, Bk, rgb_acc);
        H3D_TRACE(15);
    }
    // ================= blocks from the first skip connection on (constant style only, checked by the host) =========
    // conv 0 runs into `acc` while `cur` keeps the block input; conv 1 then accumulates on top of it (x + conv1(...)) in
    // place.  Two loops instead of one loop with a branch: at a control-flow merge of the two block kinds hipcc keeps a
    // third accumulator set alive (336 instead of 224 registers at width 448).
#pragma unroll 1
    for (int blk = A.first_skip; blk < D.n_blocks; ++blk) {
        const h3d_block_desc& Bk = D.block[blk];
        const Block<NTF, NX, T, P> K = block_view();
        f32x16 acc[NU];
        H3D_TRACE(20);
        K.prefetch_conv(Bk.spade[0]);
        K.store_const(cur, Bk.spade[0]);
        H3D_TRACE(21);
        __syncthreads();
        H3D_TRACE(22);
        K.template conv<false>(acc, Bk.spade[0]);
        H3D_TRACE(23);
        __syncthreads();                      // every wave finished reading actT
        H3D_TRACE(24);
        K.prefetch_conv(Bk.spade[1]);
        K.store_const(acc, Bk.spade[1]);
        H3D_TRACE(25);
        __syncthreads();
        H3D_TRACE(26);
        K.template conv<true>(cur, Bk.spade[1]);
        H3D_TRACE(27);
        __syncthreads();
        H3D_TRACE(28);
        if (Bk.to_rgb) K.to_rgb(cur, Bk, rgb_acc);
        H3D_TRACE(29);
    }
    if (t < 192) {
        const int c = t >> 6, pm = t & 63;
        const int64_t p = p0 + pm;
        if (p < HW) A.rgb[((int64_t)b * 3 + c) * HW + p] = rgb_acc;
    }
    H3D_TRACE_DUMP(A.rgb);         // development builds only: the trace of workgroup (1000, 3) overwrites the head of the image
    if constexpr (P == 4) {
        // sticky range flag of the x2 tier, per sample: |activation| >= 2^15 (or non-finite) somewhere in this sample's tiles
        if (A.ovf && !(gmax < 32768.f)) atomicOr(A.ovf + b, 1);
    }
}

size_t lds_bytes(int NT) {
    return (size_t)2 * (2 * NT) * 2048 + 2 * kKSA * 2048 + sizeof(float) * (4 * 3 * 64 + 64 + 64 + 256 + 128);
}

template <int NTF, int NX, typename T, int P>
int launch_one(const Args& A, int B, int64_t tiles, hipStream_t st) {
    H3D_ALLOW_MAX_LDS((synthesis_x3t_kernel<NTF, NX, T, P>));
    h3d::pre_launch();
    hipLaunchKernelGGL((synthesis_x3t_kernel<NTF, NX, T, P>), dim3((unsigned)tiles, (unsigned)B), dim3(256), lds_bytes(A.NT), st, A);
    return h3d::launch_status("h3d_synthesis_x3t");
}

template <typename T, int P>
int launch(const Args& A, int B, int64_t tiles, hipStream_t st) {
    switch (A.NT) {
        case 4: return launch_one<1, 0, T, P>(A, B, tiles, st);
        case 6: return launch_one<1, 1, T, P>(A, B, tiles, st);
        case 8: return launch_one<2, 0, T, P>(A, B, tiles, st);
        case 10: return launch_one<2, 1, T, P>(A, B, tiles, st);
        case 12: return launch_one<3, 0, T, P>(A, B, tiles, st);
        case 14: return launch_one<3, 1, T, P>(A, B, tiles, st);
        default:
            h3d::set_error("h3d_synthesis_x3t: unsupported tile count %d", A.NT);
            return H3D_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" int h3d_synthesis_x3t_tiles(int C) {
    if (C < 1 || C > 448) return -1;
    int nt = (C + 31) / 32;
    if (nt < 4) nt = 4;
    return nt + (nt & 1);
}

static int synthesis_x3t_tier(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G,
                              int g_channels, int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab,
                              float* rgb, int B, int H, int W, int dtype, int products, h3d_stream_t stream, int* ovf,
                              const int* run_if) {
    H3D_REQUIRE(wblob && tables && desc && rgb, "h3d_synthesis_x3t: null pointer");
    H3D_REQUIRE((dtype == 0 && products == 3) || (dtype == 1 && (products == 1 || products == 2 || products == 4)),
                "h3d_synthesis_x3t_tier: (dtype, products) must be (0 bf16, 3), (1 f16, 2), (1 f16, 1) or (1 f16, 4 = x2: f16 hi fragments "
                "+ fp6 records)");
    H3D_REQUIRE(h3d::aligned16(wblob) && h3d::aligned16(tables), "h3d_synthesis_x3t: weights / tables must be 16-byte aligned");
    H3D_REQUIRE(desc->n_blocks >= 1 && desc->n_blocks <= H3D_MAX_BLOCKS, "h3d_synthesis_x3t: n_blocks=%d", desc->n_blocks);
    H3D_REQUIRE(B >= 0 && B <= 65535 && H >= 1 && W >= 1, "h3d_synthesis_x3t: bad output shape");
    const int NT = h3d_synthesis_x3t_tiles(desc->C);
    if (NT < 0) {
        h3d::set_error("h3d_synthesis_x3t: width %d exceeds the 448 its LDS tile holds (use h3d_synthesis)", desc->C);
        return H3D_EUNSUPPORTED;
    }
    bool any_pixel = false, any_const = false;
    int first_skip = desc->n_blocks;
    for (int k = 0; k < desc->n_blocks; ++k) {
        if (desc->block[k].skip && first_skip == desc->n_blocks) first_skip = k;
        if (k > first_skip && !desc->block[k].skip) {
            h3d::set_error("h3d_synthesis_x3t: a block without skip connection after the first skip block is not supported; "
                           "use h3d_synthesis");
            return H3D_EUNSUPPORTED;
        }
    }
    for (int k = 0; k < desc->n_blocks; ++k)
        for (int s = 0; s < 2; ++s) {
            const h3d_spade_desc& sp = desc->block[k].spade[s];
            H3D_REQUIRE((sp.w_conv & 15) == 0 && (sp.b_conv & 3) == 0, "h3d_synthesis_x3t: misaligned conv offsets");
            if (sp.pixel_style) {
                any_pixel = true;
                if (desc->block[k].skip) {
                    h3d::set_error("h3d_synthesis_x3t: a per-pixel-style SPADE inside a skip block needs a third accumulator "
                                   "set; use h3d_synthesis");
                    return H3D_EUNSUPPORTED;
                }
                H3D_REQUIRE(sp.g_offset >= 0 && sp.g_offset + kShared <= g_channels && (sp.g_offset & 3) == 0,
                            "h3d_synthesis_x3t: block %d spade %d g_offset out of range", k, s);
                H3D_REQUIRE(sp.cst_index >= 0 && sp.cst_index < n_cst, "h3d_synthesis_x3t: cst_index out of range");
                H3D_REQUIRE((sp.w_gamma & 15) == 0 && (sp.w_beta & 15) == 0 && (sp.vec & 3) == 0, "h3d_synthesis_x3t: misaligned offsets");
            } else {
                any_const = true;
                H3D_REQUIRE(sp.ab_index >= 0 && sp.ab_index < n_ab, "h3d_synthesis_x3t: ab_index out of range");
            }
        }
    H3D_REQUIRE(!any_pixel || (G && cst && Hr >= 1 && Wr >= 1 && (g_channels & 3) == 0 && h3d::aligned16(G) && h3d::aligned16(cst)),
                "h3d_synthesis_x3t: per-pixel style blocks need G/cst (16-byte aligned, channels %% 4 == 0)");
    H3D_REQUIRE(!any_const || (ab && h3d::aligned16(ab)), "h3d_synthesis_x3t: constant-style blocks need the ab table");
    H3D_REQUIRE(desc->block[desc->n_blocks - 1].to_rgb, "h3d_synthesis_x3t: the last block must feed ToRGB");
    if (B == 0) return H3D_OK;
    Args A{};
    A.wblob = static_cast<const unsigned char*>(wblob);
    A.tables = tables;
    A.D = *desc;
    A.G = G; A.cst = cst; A.ab = ab; A.rgb = rgb;
    A.g_channels = g_channels; A.Hr = Hr; A.Wr = Wr; A.n_cst = n_cst; A.n_ab = n_ab; A.H = H; A.W = W;
    A.NT = NT;
    A.first_skip = first_skip;
    A.ovf = ovf; A.run_if = run_if;
    const int64_t tiles = ((int64_t)H * W + 63) / 64;
    H3D_REQUIRE(tiles < (int64_t(1) << 31), "h3d_synthesis_x3t: image too large");
    H3D_REQUIRE(lds_bytes(NT) <= 160 * 1024, "h3d_synthesis_x3t: width %d does not fit the 160 KB LDS", desc->C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 0) return launch<BF16, 3>(A, B, tiles, st);
    return products == 4 ? launch<F16, 4>(A, B, tiles, st) : products == 2 ? launch<F16, 2>(A, B, tiles, st) : launch<F16, 1>(A, B, tiles, st);
}

extern "C" int h3d_synthesis_x3t_tier(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G,
                                      int g_channels, int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab,
                                      float* rgb, int B, int H, int W, int dtype, int products, h3d_stream_t stream) {
    return synthesis_x3t_tier(wblob, tables, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W, dtype, products, stream,
                              nullptr, nullptr);
}

/* Range-guarded pair of the LDS-resident engine (round 4; see h3d_synthesis_x2_guarded): products == 4 (the x2 tier) ORs 1
 * into *flag when an activation it converted was >= 2^15 in magnitude or non-finite; any other tier (use (0, 3): bf16 planes,
 * fp32 exponent range) returns at once, leaving rgb untouched, when *flag == 0. */
extern "C" int h3d_synthesis_x3t_tier_guarded(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G,
                                              int g_channels, int Hr, int Wr, const float* cst, int n_cst, const float* ab,
                                              int n_ab, float* rgb, int B, int H, int W, int dtype, int products, int* flag,
                                              h3d_stream_t stream) {
    H3D_REQUIRE(flag, "h3d_synthesis_x3t_tier_guarded: null flag");
    return synthesis_x3t_tier(wblob, tables, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W, dtype, products, stream,
                              products == 4 ? flag : nullptr, products == 4 ? nullptr : flag);
}

extern "C" int h3d_synthesis_x3t(const void* wblob, const float* tables, const h3d_synth_desc* desc, const float* G,
                                 int g_channels, int Hr, int Wr, const float* cst, int n_cst, const float* ab, int n_ab,
                                 float* rgb, int B, int H, int W, h3d_stream_t stream) {
    return h3d_synthesis_x3t_tier(wblob, tables, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W, 0, 3, stream);
}
