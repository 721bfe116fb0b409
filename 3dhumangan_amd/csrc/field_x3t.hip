// A5 (+A6 fused) on the f16 matrix cores with split ("x3") operands and LDS-resident activations ("x3t"): the FiLM-SIREN
// field for hidden widths the register-resident engine (field_x3.hip, <= 256) cannot hold -- MAP3DBN (384) and
// MAP3DBN512L (420) -- and, being width-generic, any width up to 448.  gfx950 only.
//
// Reference semantics: lib/implicit_funcitions/modulated.py:41-75, lib/components/pigan_layers.py:63-87,
// lib/generators/volume_rendering.py:12-56 (same as neural_field.hip / field_x3.hip; engine: x3t_common.hpp).
//
// One 256-thread workgroup walks max(64, S) samples (whole rays when fused) in tiles of 64.  Per tile: inputs staged
// as B fragments; nine GEMMs (coordinate layer K=3, FiLM-0 coordinate half, geometry layer K=31, FiLM-0 geometry half,
// FiLM 1-3, colour layer + one k-step for the view direction, feature head) with the FiLM sine epilogue
// y = v_sin(acc*A1 + A0) (affine, bias, de-scaling and 1/2pi folded into two per-channel LDS tables) writing the next
// layer's fragments; density / colour heads as fp32 dot products over the fragments; in the fused kernel wave 0 turns
// the 64 densities into compositing weights (segmented wavefront product scan) and the feature head runs with swapped
// operands so that the sum over a ray's samples is a sum over accumulator registers -- the [N, F+4] field tensor never
// exists in HBM.  MFMA-bound: 2*(7*Hd^2 + 41*Hd) flop per sample (x3 products issued).
#include "x3t_common.hpp"
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

using namespace h3d;

namespace {

typedef F16::vec8 half8;

enum { ST_COORD = 0, ST_GEO, ST_FILM0, ST_FILM1, ST_FILM2, ST_FILM3, ST_COLOR, ST_COUNT };
enum { W_COORD = 0, W_GEO, W_F0, W_F1, W_F2, W_F3, W_COLOR, W_FEAT, W_COUNT };
enum { IN_COORD = 0, IN_GEO = 1, IN_DIR = 3, IN_SLOTS = 4 };      // k-step slots of the input tile

constexpr float kSInT = 64.f;      // input scale (coords / geometry features / view direction), as field_x3.hip

struct LayoutT {           // offsets in BYTES into the blob (all multiples of 16)
    int NT, KS, HdP;
    int64_t w[W_COUNT];
    int64_t inv_scale;     // float[W_COUNT]   1 / (weight scale * input scale)
    int64_t bias;          // float[ST_COUNT][HdP]
    int64_t b_feat;        // float[HdP]
    int64_t head_w;        // f16 [KS][hi|lo][64 lanes][8]: A fragments of the head tile, rows 0..3 = sigma, r, g, b (each scaled
                           //   by its own power of two), rows 4..31 zero; K in accumulator order
    int64_t head_inv;      // float[4]   1 / head scale
    int64_t head_b;        // float[4]
    int64_t total;
};

int kstot_of(int wi, int KS) { return wi == W_COORD ? 1 : wi == W_GEO ? 2 : wi == W_F0 ? 2 * KS : wi == W_COLOR ? KS + 1 : KS; }
// bytes of one tile of matrix wi: the two input layers always travel in the x3 format; with `x2c` (the x2 tier's blob) the
// accumulator-fed matrices are in the x2c format (x3t_common.hpp: 3 KiB per K-tile, the colour layer's trailing view-direction
// k-step in the x3 format)
__host__ __device__ inline int64_t tile_bytes_of(int wi, int KS, bool x2c) {
    const int n = wi == W_COORD ? 1 : wi == W_GEO ? 2 : wi == W_F0 ? 2 * KS : wi == W_COLOR ? KS + 1 : KS;
    return (x2c && wi >= W_F0) ? x3t_tile_bytes<true>(n) : x3t_tile_bytes<false>(n);
}

LayoutT make_layout(int Hd, int F) {       // (one layout for both blob kinds: the x2c tiles use the first three quarters of their regions)
    LayoutT L;
    const int w = Hd > F ? Hd : F;
    int nt = (w + 31) / 32;
    if (nt < 4) nt = 4;
    nt += nt & 1;                       // even tile counts only: 4*NTF or 4*NTF + 2
    L.NT = nt;
    L.KS = 2 * nt;
    L.HdP = 32 * nt;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 15) / 16 * 16; return r; };
    for (int i = 0; i < W_COUNT; ++i) L.w[i] = take((int64_t)L.NT * kstot_of(i, L.KS) * 2048);
    L.inv_scale = take(4 * W_COUNT);
    L.bias = take(4 * (int64_t)ST_COUNT * L.HdP);
    L.b_feat = take(4 * (int64_t)L.HdP);
    L.head_w = take((int64_t)L.KS * 2048);
    L.head_inv = take(16);
    L.head_b = take(16);
    L.total = o;
    return L;
}

struct Args {
    const unsigned char* blob;
    const float* points;
    const float* geo;
    const float* dirs;
    const float* freq;
    const float* phase;
    float* out;
    const float* z_vals;
    const float* noise;
    float* feats;
    float* depth;
    float* weights;
    int64_t N;
    int Hd, F, geo_stride, S, clamp_mode, last_back, white_back;
    int n_groups;          // sample groups (64 samples, or one ray when S > 64) per batch item; a workgroup walks blockIdx.x, + gridDim.x, ..
    float input_scaler;
    LayoutT L;
};

__device__ __forceinline__ float density(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? x : log1pf(expf(x));
    return fmaxf(x, 0.f);
}

// P: partial products per operand pair of the hidden GEMMs (x3t_common.hpp): 3 = fp32-class (the default engine), 1 = plain
// f16 matrix-core arithmetic (the "f16 MFMA" tier of BASELINE config 5; the K=3 / K=31 input layers always run with 3).
// P = 4: the x2 arithmetic (x3_common.hpp / x3t_common.hpp: one f16 product + one block-scaled fp6 product per contraction; the
// blob must come from h3d_field_pack_x2t).
template <int NTF, int NX, bool FUSED, int P>
__global__ __launch_bounds__(256, 1) void field_x3t_kernel(Args A) {
    constexpr int NU = 2 * NTF + NX;
    constexpr bool LO = P == 3;            // activations carry a lo half
    constexpr bool X2 = P == 4;            // activations carry the K-tile's fp6 record in the lo planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const LayoutT& L = A.L;
    const int KS = L.KS, HdP = L.HdP;
    const int act_stride = KS * 2048;                                 // bytes between the sample tiles of actT
    unsigned char* actT = smem_raw;                                    // [2][KS][2][1 KB]
    unsigned char* inT = actT + 2 * act_stride;                        // [2][IN_SLOTS][2][1 KB]
    constexpr int in_stride = IN_SLOTS * 2048;
    float* tab = reinterpret_cast<float*>(inT + 2 * in_stride);        // [ST_COUNT][2][HdP]: A1 row, A0 row per step
    float* part = tab + ST_COUNT * 2 * HdP;                            // [4 waves][4 heads][64] partial head sums
    float* wgt = part + 4 * 4 * 64;                                    // [64] compositing weights
    float* bgl = wgt + 64;                                             // [64] background term of the row's ray
    float* rgbv = bgl + 64;                                            // [64][3]
    float* xsum = rgbv + 64 * 3;                                       // [4 waves][32] extra-unit ray sums

    const int t = threadIdx.x, lane0 = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane0 & 31, h = lane0 >> 5;
    const int b = blockIdx.y;
    const int64_t N = A.N;
    const int Hd = A.Hd, F = A.F, S = A.S;
    const unsigned char* __restrict__ blob = A.blob;
    const float* __restrict__ invs = reinterpret_cast<const float*>(blob + L.inv_scale);
    const float* __restrict__ bfeat = reinterpret_cast<const float*>(blob + L.b_feat);
    const unsigned char* __restrict__ headw = blob + L.head_w;
    const float* __restrict__ headinv = reinterpret_cast<const float*>(blob + L.head_inv);
    const float* __restrict__ headb = reinterpret_cast<const float*>(blob + L.head_b);
    X3tUnits<NTF, NX> U0;
    U0.init(wave);

    // ---- per-sample-of-the-batch activation tables, once per workgroup:
    //      y = sin(f * (acc*inv + bias) + p) = v_sin(acc * A1 + A0),  A1 = inv*f/2pi,  A0 = (bias*f + p)/2pi
    {
        const float* __restrict__ bias = reinterpret_cast<const float*>(blob + L.bias);
        const float* __restrict__ fr = A.freq + (int64_t)b * 4 * Hd;
        const float* __restrict__ ph = A.phase + (int64_t)b * 4 * Hd;
        const float inv2pi = 0.15915494309189535f;
        for (int idx = t; idx < ST_COUNT * HdP; idx += 256) {
            const int st = idx / HdP, n = idx - st * HdP;
            float a1 = 0.f, a0 = 0.f;                   // padding channels: sin(0) = 0
            if (n < Hd) {
                float ff = 30.f, pp = 0.f;
                if (st >= ST_FILM0) {
                    const int sl = st == ST_COLOR ? 3 : st - ST_FILM0;
                    ff = fr[sl * Hd + n] * 15.f + 30.f;
                    pp = ph[sl * Hd + n];
                }
                const int wi = st == ST_COORD ? W_COORD : st == ST_GEO ? W_GEO : st == ST_COLOR ? W_COLOR : W_F0 + (st - ST_FILM0);
                a1 = invs[wi] * ff * inv2pi;
                a0 = fmaf(bias[st * HdP + n], ff, pp) * inv2pi;
            }
            tab[(st * 2 + 0) * HdP + n] = a1;
            tab[(st * 2 + 1) * HdP + n] = a0;
        }
    }

    H3D_TRACE_INIT();
    H3D_TRACE(0);
    const int group_pts = FUSED ? (S > 64 ? S : 64) : 64;
    const int tiles = group_pts / 64;
    const int seglen = FUSED ? (S < 64 ? S : 64) : 64;
    const float inv_f = invs[W_FEAT];
    SplitF16 split;
    // Persistent workgroups (round 6): the batch item's activation tables above are built once, then the workgroup walks the
    // sample groups blockIdx.x, blockIdx.x + gridDim.x, .. (rounds 2-5 launched one workgroup per 64 samples: 147 456 table
    // builds of 7 x HdP entries per launch at the bench size)
    int n_groups = A.n_groups;
    asm volatile("" : "+s"(n_groups));
#pragma unroll 1
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t g0 = (int64_t)grp * group_pts;
    // state carried across the tiles of a multi-tile ray (wave 0 lanes hold identical copies)
    float carryT = 1.f, carryW = 0.f, carryD = 0.f, rgbacc = 0.f;
    float rayacc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) rayacc[u] = 0.f;

    for (int ti = 0; ti < tiles; ++ti) {
        const int64_t n0 = g0 + (int64_t)ti * 64;
        const bool last_tile = ti == tiles - 1;
        // The weights do not depend on the tile: launder an opaque zero offset so that the compiler does not hoist the
        // first k-steps' fragment loads of every GEMM out of the tile loop (LICM) and spill hundreds of registers.
        // (The same goes for every per-phase fragment address derived from the lane and the wave's tiles.)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const unsigned char* wblob = blob + opaque;
#ifdef H3D_EXPERIMENT_ALIAS_W        // timing experiment (wrong results): every hidden matrix reads FiLM 1's bytes (0.8 MB: L2-resident)
        auto wmat = [&](int wi) { return wblob + L.w[wi >= W_F0 ? W_F1 : wi]; };
#else
        auto wmat = [&](int wi) { return wblob + L.w[wi]; };
#endif
        // tile stride of matrix wi and byte offset of its k-step ks0 (a K-tile boundary, or the colour layer's trailing k-step)
        auto wstride = [&](int wi) { return tile_bytes_of(wi, KS, X2); };
        auto woff = [&](int wi, int ks0) { return (X2 && wi >= W_F0) ? x3t_kstep_off<true>(ks0) : x3t_kstep_off<false>(ks0); };
        X3tUnits<NTF, NX> U = U0;
#pragma unroll
        for (int i = 0; i < NTF + NX; ++i) asm volatile("" : "+s"(U.nt[i]));
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        // FiLM epilogue of an accumulator set -> fragments of actT
        auto store_film = [&](f32x16 (&acc)[NU], int st) __attribute__((always_inline)) {
            // fresh copies of the lane id / lane half: the fragment and table addresses below are the same in every layer
            // and would otherwise be kept alive (dozens of registers) across the GEMMs instead of being recomputed
            int lane = lane0, h = lane0 >> 5;
            asm volatile("" : "+v"(lane), "+v"(h));
            const float* a1 = tab + (st * 2 + 0) * HdP + 4 * h;
            const float* a0 = tab + (st * 2 + 1) * HdP + 4 * h;
            // the two units of a tile (sample tiles 0 / 1) share the tile's table values: one fetch per tile, one tile ahead
            f32x4 t1[2][4], t0[2][4];
            auto fetch = [&](int slot, int nt) __attribute__((always_inline)) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    t1[slot][rg] = ld4(a1 + nt * 32 + rg * 8);
                    t0[slot][rg] = ld4(a0 + nt * 32 + rg * 8);
                }
            };
            fetch(0, U.nt[0]);
            static_for<0, NTF + NX>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, sl = i & 1;
                if constexpr (i + 1 < NTF + NX) fetch(sl ^ 1, U.nt[i + 1]);
                auto film = [&](int rg, f32x4 v) {
                    f32x4 y;
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[q] = __builtin_amdgcn_sinf(fmaf(v[q], t1[sl][rg][q], t0[sl][rg][q]));
                    return y;
                };
                auto store = [&](const f32x16& v, int mt) __attribute__((always_inline)) {
                    if constexpr (X2) x3t_store_unit_x2<false>(v, actT, KS, U.nt[i], mt, lane, film);      // sines: static scale
                    else x3t_store_unit<LO>(v, actT, KS, U.nt[i], mt, lane, split, film);
                };
                if constexpr (i < NTF) {
                    pin1(acc[2 * i]);
                    store(acc[2 * i], 0);
                    pin1(acc[2 * i + 1]);
                    store(acc[2 * i + 1], 1);
                } else {
                    pin1(acc[2 * NTF]);
                    store(acc[2 * NTF], U.xmt);
                }
                // bound the scheduler's hoisting to one tile (all tiles at once cost > 200 registers)
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // The four 1-row heads (density, r, g, b) of the activations in actT on the matrix cores: D[head][sample] with the
        // head tile's A fragments (4 real rows) from L2; wave w contracts k-steps w, w+4, .. and leaves its partial sums in
        // part[w][head][sample] (rows 0..3 of the tile are registers 0..3 of the lanes with h == 0).  All weight fragments
        // of the wave are requested up front (<= 7 k-steps), so their latency overlaps instead of adding up.
        auto heads = [&]() __attribute__((always_inline)) {
            if constexpr (X2) {
                // x2: wave w contracts K-tiles w, w+4, .. (k-steps 2T, 2T+1): two f16 instructions + one fp6 instruction per
                // K-tile and sample tile; the head tile's "lo" planes hold its record halves like every other matrix
                constexpr int MAXT = 4;                                 // KS <= 28 -> 14 K-tiles
                u32x4 wh[MAXT][2], wr[MAXT][2];
                const unsigned char* wsrc = headw + opaque + lane * 16;
#pragma unroll
                for (int i = 0; i < MAXT; ++i) {
                    const int T = wave + 4 * i;
                    if (2 * T < KS) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            wh[i][j] = *reinterpret_cast<const u32x4*>(wsrc + (2 * T + j) * 2048);
                            wr[i][j] = *reinterpret_cast<const u32x4*>(wsrc + (2 * T + j) * 2048 + 1024);
                        }
                    }
                }
                f32x16 ha[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ha[mt][r] = 0.f;
#pragma unroll
                for (int i = 0; i < MAXT; ++i) {
                    const int T = wave + 4 * i;
                    if (2 * T < KS) {
                        const i32x8 w6 = {(int)wr[i][0][0], (int)wr[i][0][1], (int)wr[i][0][2], (int)wr[i][0][3],
                                          (int)wr[i][1][0], (int)wr[i][1][1], (int)wr[i][1][2], (int)wr[i][1][3]};
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            u32x4 xr[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const unsigned char* q = actT + x3t_frag(KS, mt, 2 * T + j, 0) + lane * 16;
                                const half8 xh_ = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(q));
                                xr[j] = *reinterpret_cast<const u32x4*>(q + 1024);
                                ha[mt] = F16::mfma(__builtin_bit_cast(half8, wh[i][j]), xh_, ha[mt]);
                            }
                            const i32x8 x6 = {(int)xr[0][0], (int)xr[0][1], (int)xr[0][2], (int)xr[0][3],
                                              (int)xr[1][0], (int)xr[1][1], (int)xr[1][2], (int)xr[1][3]};
                            ha[mt] = mm6<false>(w6, x6, ha[mt]);
                        }
                    }
                }
                if (h == 0) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int hd = 0; hd < 4; ++hd) part[(wave * 4 + hd) * 64 + mt * 32 + m] = ha[mt][hd];
                }
                return;
            }
            constexpr int MAXK = 7;                                     // KS <= 28
            u32x4 wh[MAXK], wl[MAXK];
            const unsigned char* wsrc = headw + opaque + lane * 16;
#pragma unroll
            for (int i = 0; i < MAXK; ++i) {
                const int ks = wave + 4 * i;
                if (ks < KS) {
                    wh[i] = *reinterpret_cast<const u32x4*>(wsrc + ks * 2048);
                    if (LO) wl[i] = *reinterpret_cast<const u32x4*>(wsrc + ks * 2048 + 1024);
                }
            }
            f32x16 ha[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ha[mt][r] = 0.f;
#pragma unroll
            for (int i = 0; i < MAXK; ++i) {
                const int ks = wave + 4 * i;
                if (ks < KS) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const unsigned char* q = actT + x3t_frag(KS, mt, ks, 0) + lane * 16;
                        const half8 xh_ = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(q));
                        ha[mt] = F16::mfma(__builtin_bit_cast(half8, wh[i]), xh_, ha[mt]);
                        if (LO) {
                            const half8 xl_ = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(q + 1024));
                            ha[mt] = F16::mfma(__builtin_bit_cast(half8, wl[i]), xh_, ha[mt]);
                            ha[mt] = F16::mfma(__builtin_bit_cast(half8, wh[i]), xl_, ha[mt]);
                        }
                    }
                }
            }
            if (h == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int hd = 0; hd < 4; ++hd) part[(wave * 4 + hd) * 64 + mt * 32 + m] = ha[mt][hd];
            }
        };
        auto head_value = [&](int hd, int sample) {                    // after a barrier: sum over the waves, de-scale, bias
            const float* q = part + hd * 64 + sample;
            return ((q[0] + q[256]) + (q[512] + q[768])) * headinv[hd] + headb[hd];
        };
        auto zero = [&](f32x16 (&acc)[NU]) __attribute__((always_inline)) {
    #pragma unroll
            for (int u = 0; u < NU; ++u)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        };


        // ---- stage the inputs as B fragments (natural K order): thread -> (sample tile, slot lane, k-step)
        {
            const int sl = t & 63, q = t >> 6;                           // q: 0..3
            const int sm = sl & 31, sh = sl >> 5;
            // coordinates (q = 0, 1 -> mt) and view direction (q = 2, 3 -> mt): k = 8*sh + e < 3
            {
                const int mt = q & 1;
                const bool is_dir = q >= 2;
                const int64_t n = n0 + mt * 32 + sm;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (sh == 0) {
                    if (!is_dir) {
                        if (n < N) {
                            const float* p = A.points + ((int64_t)b * N + n) * 3;
                            v[0] = p[0] * A.input_scaler * kSInT; v[1] = p[1] * A.input_scaler * kSInT; v[2] = p[2] * A.input_scaler * kSInT;
                        }
                    } else if (A.dirs) {                // unit vectors: no scaling needed (f16 subnormals are honoured)
                        if (n < N) {
                            const float* p = A.dirs + ((int64_t)b * N + n) * 3;
                            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
                        }
                    } else {
                        v[2] = -1.f;                   // lock_view_dependence: (0, 0, -1)
                    }
                }
                u32x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; e += 2) { unsigned l2; hi[e / 2] = split(v[e], v[e + 1], l2); lo[e / 2] = l2; }
                unsigned char* dst = inT + mt * in_stride + (is_dir ? IN_DIR : IN_COORD) * 2048 + sl * 16;
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + 1024) = lo;
            }
            // geometry features: (q & 1) -> mt, (q >> 1) -> k-step; k = 16*ks + 8*sh + e < 31
            {
                const int mt = q & 1, ks = q >> 1;
                const int64_t n = n0 + mt * 32 + sm;
                const float* g = A.geo + ((int64_t)b * N + (n < N ? n : N - 1)) * A.geo_stride;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 16 + sh * 8 + e;
                    v[e] = (k < 31 && n < N) ? g[k < 31 ? k : 0] * kSInT : 0.f;
                }
                u32x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; e += 2) { unsigned l2; hi[e / 2] = split(v[e], v[e + 1], l2); lo[e / 2] = l2; }
                unsigned char* dst = inT + mt * in_stride + (IN_GEO + ks) * 2048 + sl * 16;
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + 1024) = lo;
            }
        }
        __syncthreads();          // inputs (and, first tile, the tables) visible; previous tile's readers are done
        H3D_TRACE(1);

        f32x16 acc[NU], acc2[NU];
        X3tRing<NTF + NX> ring;     // weight fragments in flight; the next GEMM's first k-steps are requested before the
                                    // epilogue and barriers in front of it (x3t_prefetch)
        // ---- coordinate layer (K = 3) -> sine -> FiLM 0, coordinate half
        zero(acc);
        gemm_x3t<F16, NTF, NX, false, true>(acc, inT + IN_COORD * 2048, in_stride, wmat(W_COORD), wstride(W_COORD), 0, 1, U, lane, ring);
        x3t_prefetch<NTF, NX, P>(ring, wmat(W_F0), wstride(W_F0), 0, U, lane);
        H3D_TRACE(2);
        store_film(acc, ST_COORD);
        H3D_TRACE(3);
        __syncthreads();
        H3D_TRACE(4);
        zero(acc2);
        gemm_x3t<F16, NTF, NX, false, false, true, P>(acc2, actT, act_stride, wmat(W_F0), wstride(W_F0), 0, KS, U, lane, ring);
        H3D_TRACE(5);
        // ---- geometry layer (K = 31) -> sine -> FiLM 0, geometry half (same accumulators)
        zero(acc);
        gemm_x3t<F16, NTF, NX, false, true>(acc, inT + IN_GEO * 2048, in_stride, wmat(W_GEO), wstride(W_GEO), 0, 2, U, lane, ring);
        x3t_prefetch<NTF, NX, P>(ring, wmat(W_F0), wstride(W_F0), woff(W_F0, KS), U, lane);
        H3D_TRACE(6);
        __syncthreads();          // every wave has finished reading the coordinate activations
        H3D_TRACE(7);
        store_film(acc, ST_GEO);
        H3D_TRACE(8);
        __syncthreads();
        H3D_TRACE(9);
        gemm_x3t<F16, NTF, NX, false, false, true, P>(acc2, actT, act_stride, wmat(W_F0), wstride(W_F0), woff(W_F0, KS), KS, U, lane, ring);
        x3t_prefetch<NTF, NX, P>(ring, wmat(W_F1), wstride(W_F1), 0, U, lane);
        H3D_TRACE(10);
        __syncthreads();
        H3D_TRACE(11);
        store_film(acc2, ST_FILM0);
        H3D_TRACE(12);
        __syncthreads();
        H3D_TRACE(13);
        // ---- FiLM 1..3
#pragma unroll 1
        for (int l = 1; l < 4; ++l) {
            zero(acc);
            gemm_x3t<F16, NTF, NX, false, false, true, P>(acc, actT, act_stride, wmat(W_F0 + l), wstride(W_F0 + l), 0, KS, U, lane, ring);
            x3t_prefetch<NTF, NX, P>(ring, wmat(W_F0 + l + 1), wstride(W_F0 + l + 1), 0, U, lane);      // FiLM l+1, or the colour layer
            H3D_TRACE(14);
            __syncthreads();
            H3D_TRACE(15);
            store_film(acc, ST_FILM0 + l);
            H3D_TRACE(16);
            __syncthreads();
            H3D_TRACE(17);
        }

        // ---- density head (and, unused here, the colour heads of the same tile) on the matrix cores
        heads();
        H3D_TRACE(18);
        __syncthreads();
        H3D_TRACE(19);
        if (t < 64) {
            const float sigma = head_value(0, t);
            const int64_t n = n0 + t;
            const bool ok = n < N;
            if (!FUSED) {
                if (ok) A.out[((int64_t)b * N + n) * (F + 4) + F + 3] = sigma;
            } else {
                // ---- compositing weights of the 64 samples of this tile (volume_rendering.py:18-46)
                const int s_idx = (int)(n % S);
                const int64_t gi = (int64_t)b * N + n;
                float alpha = 0.f, f = 1.f, z = 0.f;
                if (ok) {
                    z = A.z_vals[gi];
                    const float delta = (s_idx == S - 1) ? 1e9f : A.z_vals[gi + 1] - z;
                    const float sg = sigma + (A.noise ? A.noise[gi] : 0.f);
                    alpha = 1.f - expf(-delta * density(sg, A.clamp_mode));
                    f = (1.f - alpha) + 1e-12f;
                }
                const int sl = t & (seglen - 1);
                float incl = f;
                for (int off = 1; off < seglen; off <<= 1) {
                    const float u = __shfl_up(incl, off, 64);
                    if (sl >= off) incl *= u;
                }
                float excl = __shfl_up(incl, 1, 64);
                if (sl == 0) excl = 1.f;
                float w = alpha * (carryT * excl);
                float wsum = w, dsum = w * z;
                for (int off = seglen >> 1; off > 0; off >>= 1) {
                    wsum += __shfl_xor(wsum, off, 64);
                    dsum += __shfl_xor(dsum, off, 64);
                }
                const float z_last = __shfl(z, t | (seglen - 1), 64);
                carryT *= __shfl(incl, 63, 64);
                carryW += wsum;
                carryD += dsum;
                float bg = 0.f;
                if (last_tile) {
                    bg = 1.f - carryW;
                    if (ok && s_idx == S - 1) {
                        A.depth[gi / S] = carryD + bg * z_last;
                        if (A.last_back) w += bg;
                    }
                }
                if (ok) A.weights[gi] = w;
                wgt[t] = w;
                bgl[t] = bg;
            }
        }

        H3D_TRACE(20);
        // ---- colour FiLM on [x, dir]: KS k-steps over the hidden features + one k-step carrying the view direction
        zero(acc);
        gemm_x3t<F16, NTF, NX, false, false, true, P>(acc, actT, act_stride, wmat(W_COLOR), wstride(W_COLOR), 0, KS, U, lane, ring);
        gemm_x3t<F16, NTF, NX, false, true>(acc, inT + IN_DIR * 2048, in_stride, wmat(W_COLOR), wstride(W_COLOR), woff(W_COLOR, KS), 1, U, lane, ring);
        x3t_prefetch<NTF, NX, P>(ring, wmat(W_FEAT), wstride(W_FEAT), 0, U, lane);
        H3D_TRACE(21);
        __syncthreads();
        H3D_TRACE(22);
        store_film(acc, ST_COLOR);
        H3D_TRACE(23);
        __syncthreads();
        H3D_TRACE(24);

        // ---- colour heads (rows 1..3 of the head tile) and feature head (operands swapped: rows = samples)
        heads();
        H3D_TRACE(25);
        f32x16 (&accF)[NU] = acc2;
        zero(accF);
        gemm_x3t<F16, NTF, NX, true, false, true, P>(accF, actT, act_stride, wmat(W_FEAT), wstride(W_FEAT), 0, KS, U, lane, ring);
        H3D_TRACE(26);
        __syncthreads();
        H3D_TRACE(27);
        if (t < 192) {
            const int c = t >> 6, mm_ = t & 63;
            const float v = head_value(1 + c, mm_);
            const float rgb = 1.f / (1.f + expf(-v));
            const int64_t n = n0 + mm_;
            if (!FUSED) {
                if (n < N) A.out[((int64_t)b * N + n) * (F + 4) + c] = rgb;
            } else {
                rgbv[mm_ * 3 + c] = rgb;
            }
        }
        if (!FUSED) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int n = U.tile(u) * 32 + m, mt = U.mt(u);
                if (n >= F) continue;
                const float bias = bfeat[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + (r >> 2) * 8 + 4 * h + (r & 3);
                    const int64_t pn = n0 + row;
                    if (pn < N) A.out[((int64_t)b * N + pn) * (F + 4) + 3 + n] = fmaf(accF[u][r], inv_f, bias);
                }
            }
        } else {
            __syncthreads();       // rgbv visible
            const int C = F + 3;
            const int nseg = 64 / seglen;
            // colour channels: a few threads walk their ray's rows
            if (t < nseg * 3) {
                const int seg = t / 3, c = t - seg * 3;
                float s = 0.f;
                for (int q = 0; q < seglen; ++q) s = fmaf(wgt[seg * seglen + q], rgbv[(seg * seglen + q) * 3 + c], s);
                rgbacc += s;     // only meaningful for nseg == 1 (multi-tile rays); otherwise reset each tile
                const int64_t n_first = n0 + (int64_t)seg * seglen;
                if (last_tile && n_first < N) {
                    const int64_t ray = ((int64_t)b * N + n_first) / S;
                    const float tot = (S > 64 ? rgbacc : s) + (A.white_back ? bgl[seg * seglen] : 0.f);
                    A.feats[ray * C + c] = tot;
                }
            }
            // feature channels: weighted sum over the rows of each ray, straight from the accumulators
            //   sum_r w_r * (acc_r * inv + bias)
            float xs_last = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int n = U.tile(u) * 32 + m, mt = U.mt(u);
                const bool okn = n < F;
                const float bias = okn ? bfeat[n] : 0.f;
                float s4[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 w4 = ld4(wgt + mt * 32 + rg * 8 + 4 * h);
                    float s = fmaf(accF[u][rg * 4 + 0], inv_f, bias) * w4[0];
                    s = fmaf(fmaf(accF[u][rg * 4 + 1], inv_f, bias), w4[1], s);
                    s = fmaf(fmaf(accF[u][rg * 4 + 2], inv_f, bias), w4[2], s);
                    s = fmaf(fmaf(accF[u][rg * 4 + 3], inv_f, bias), w4[3], s);
                    s += __shfl_xor(s, 32, 64);
                    s4[rg] = s;
                }
                if (S >= 64) {
                    rayacc[u] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
                    if (u < 2 * NTF) {
                        // the ray spans both sample tiles: units 2i and 2i+1 of this wave
                        if ((u & 1) && last_tile && okn && h == 0 && n0 < N) {
                            const int64_t ray = ((int64_t)b * N + n0) / S;
                            A.feats[ray * C + 3 + n] = (rayacc[u > 0 ? u - 1 : 0] + rayacc[u]) + (A.white_back ? bgl[0] : 0.f);
                        }
                    } else {
                        xs_last = rayacc[u];
                    }
                } else {
                    const int g8 = S >> 3;                       // 8-row groups per ray: 1, 2 or 4
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        if (rg % g8 != 0) continue;
                        float s = s4[rg];
                        if (g8 >= 2) s += s4[rg + 1 < 4 ? rg + 1 : 3];
                        if (g8 == 4) s += s4[2] + s4[3];
                        const int m_first = mt * 32 + rg * 8;
                        const int64_t n_first = n0 + m_first;
                        if (okn && h == 0 && n_first < N) {
                            const int64_t ray = ((int64_t)b * N + n_first) / S;
                            A.feats[ray * C + 3 + n] = s + (A.white_back ? bgl[m_first] : 0.f);
                        }
                    }
                }
            }
            if (NX && S >= 64 && last_tile) {
                // the extra tile's two sample tiles sit in different waves (w and w^1): meet in LDS
                if (h == 0) xsum[wave * 32 + m] = xs_last;
                __syncthreads();
                if ((wave & 1) == 0 && h == 0) {
                    const int n = U.nt[NTF + NX - 1] * 32 + m;
                    if (n < F && n0 < N) {
                        const int64_t ray = ((int64_t)b * N + n0) / S;
                        A.feats[ray * C + 3 + n] = (xsum[wave * 32 + m] + xsum[(wave + 1) * 32 + m]) + (A.white_back ? bgl[0] : 0.f);
                    }
                }
            }
        }
        H3D_TRACE(28);
        __syncthreads();     // actT / inT / part / wgt are rewritten by the next tile
        H3D_TRACE(29);
    }
    }   // sample groups
    H3D_TRACE_DUMP(A.out);
}

size_t lds_bytes(const LayoutT& L) {
    return (size_t)2 * L.KS * 2048 + 2 * IN_SLOTS * 2048 +
           sizeof(float) * ((size_t)ST_COUNT * 2 * L.HdP + 4 * 4 * 64 + 64 + 64 + 64 * 3 + 4 * 32);
}

template <int NTF, int NX, bool FUSED, int P>
int launch_one(Args A, int B, int64_t groups, hipStream_t st) {
    H3D_ALLOW_MAX_LDS((field_x3t_kernel<NTF, NX, FUSED, P>));
    A.n_groups = (int)groups;
    // about eight persistent workgroups per CU in total (one resident per CU: LDS): tables once per many groups, short tail
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
    }
    static const int per_cu = getenv("H3D_FIELD_X3T_WG_PER_CU") ? atoi(getenv("H3D_FIELD_X3T_WG_PER_CU")) : 8;      // 0: one group per workgroup
    const int64_t per_sample = per_cu <= 0 ? groups : std::max<int64_t>(1, std::min<int64_t>(groups, ((int64_t)per_cu * cus + B - 1) / B));
    h3d::pre_launch();
    hipLaunchKernelGGL((field_x3t_kernel<NTF, NX, FUSED, P>), dim3((unsigned)per_sample, (unsigned)B), dim3(256), lds_bytes(A.L), st, A);
    return h3d::launch_status(FUSED ? "h3d_render_fused_x3t" : "h3d_neural_field_x3t");
}

template <bool FUSED, int P>
int launch(const Args& A, int B, int64_t groups, hipStream_t st) {
    switch (A.L.NT) {
        case 4: return launch_one<1, 0, FUSED, P>(A, B, groups, st);
        case 6: return launch_one<1, 1, FUSED, P>(A, B, groups, st);
        case 8: return launch_one<2, 0, FUSED, P>(A, B, groups, st);
        case 10: return launch_one<2, 1, FUSED, P>(A, B, groups, st);
        case 12: return launch_one<3, 0, FUSED, P>(A, B, groups, st);
        case 14: return launch_one<3, 1, FUSED, P>(A, B, groups, st);
        default:
            h3d::set_error("x3t field kernel: width %d exceeds the 448 its LDS tile holds (use the fp32 engine)", A.L.HdP);
            return H3D_EUNSUPPORTED;
    }
}

float pow2_scale(const float* w, int64_t n, float target) {
    float mx = 0.f;
    for (int64_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    if (mx == 0.f) return 1.f;
    return exp2f(floorf(log2f(target / mx)));
}

bool widths_ok(int Hd, int F) { return Hd >= 1 && F >= 1 && Hd <= 448 && F <= 448; }

int check_x3t(const void* packed, const float* points, const float* geo, const float* freq, const float* phase,
              int B, int64_t N, int Hd, int F, int geo_stride) {
    H3D_REQUIRE(packed && points && geo && freq && phase, "x3t field: null pointer");
    H3D_REQUIRE(h3d::aligned16(packed), "x3t field: packed weights must be 16-byte aligned");
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "x3t field: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(Hd >= 1 && F >= 1, "x3t field: bad widths");
    H3D_REQUIRE(geo_stride >= 31, "x3t field: geo_stride=%d must be >= 31", geo_stride);
    if (!widths_ok(Hd, F)) {
        h3d::set_error("x3t field kernel: widths up to 448 (got %d/%d); use the fp32 engine", Hd, F);
        return H3D_EUNSUPPORTED;
    }
    return H3D_OK;
}

int field_pack_t(const h3d_field_params* p, int Hd, int F, void* blob_, bool x2);

}  // namespace

extern "C" int64_t h3d_field_pack_x3t_size(int Hd, int F) {
    if (!widths_ok(Hd, F)) return -1;
    return make_layout(Hd, F).total;
}

extern "C" int h3d_field_x3t_layout(int Hd, int F, int64_t* out, int n_out) {
    H3D_REQUIRE(out && n_out >= 3 + W_COUNT + 7, "h3d_field_x3t_layout: need room for %d values", 3 + W_COUNT + 7);
    H3D_REQUIRE(widths_ok(Hd, F), "h3d_field_x3t_layout: widths up to 448 (got %d, %d)", Hd, F);
    const LayoutT L = make_layout(Hd, F);
    int i = 0;
    out[i++] = L.NT; out[i++] = L.KS; out[i++] = L.HdP;
    for (int w = 0; w < W_COUNT; ++w) out[i++] = L.w[w];
    out[i++] = L.inv_scale; out[i++] = L.bias; out[i++] = L.b_feat; out[i++] = L.head_w; out[i++] = L.head_inv; out[i++] = L.head_b;
    out[i++] = L.total;
    return H3D_OK;
}

extern "C" int h3d_field_pack_x3t(const h3d_field_params* p, int Hd, int F, void* blob) { return field_pack_t(p, Hd, F, blob, false); }
/* the x2 tier's blob (products = 4 of the _tier entry points): same layout and size; the matrices fed by accumulators and the
 * head tile carry f16 hi fragments + fp6 records instead of hi + lo fragments */
extern "C" int h3d_field_pack_x2t(const h3d_field_params* p, int Hd, int F, void* blob) { return field_pack_t(p, Hd, F, blob, true); }

namespace {
int field_pack_t(const h3d_field_params* p, int Hd, int F, void* blob_, bool x2) {
    H3D_REQUIRE(p && blob_, "h3d_field_pack_x3t: null pointer");
    H3D_REQUIRE(widths_ok(Hd, F), "h3d_field_pack_x3t: widths up to 448 (got %d, %d)", Hd, F);
    const LayoutT L = make_layout(Hd, F);
    unsigned char* blob = static_cast<unsigned char*>(blob_);
    memset(blob, 0, L.total);
    float* invs = reinterpret_cast<float*>(blob + L.inv_scale);
    const float target = 8192.f;
    auto dst = [&](int wi) { return blob + L.w[wi]; };
    // accumulator-order matrices: x3 format (hi + lo fragments) or, the x2 tier's blob, x2c (hi fragments + lo records, 3 KiB per K-tile)
    auto pack_acc = [&](const float* w, int ld, int in_begin, int in_count, int n_out, int ks0, int KSm, float sc, int wi) {
        const int64_t stride = tile_bytes_of(wi, L.KS, x2);
        if (x2) x3t_pack_x2(w, ld, in_begin, in_count, n_out, L.NT, stride, x3t_kstep_off<true>(ks0), KSm, sc, dst(wi));
        else x3t_pack_f16(w, ld, in_begin, in_count, n_out, L.NT, stride, x3t_kstep_off<false>(ks0), KSm, sc, dst(wi), true);
    };
    // input layers: natural K order
    {
        const float sc = pow2_scale(p->w_coord, (int64_t)Hd * 3, target);
        x3t_pack_f16(p->w_coord, 3, 0, 3, Hd, L.NT, tile_bytes_of(W_COORD, L.KS, x2), 0, 1, sc, dst(W_COORD), false);
        invs[W_COORD] = 1.f / (sc * kSInT);
    }
    {
        const float sc = pow2_scale(p->w_geo, (int64_t)Hd * 31, target);
        x3t_pack_f16(p->w_geo, 31, 0, 31, Hd, L.NT, tile_bytes_of(W_GEO, L.KS, x2), 0, 2, sc, dst(W_GEO), false);
        invs[W_GEO] = 1.f / (sc * kSInT);
    }
    // FiLM 0: both K halves accumulate into the same registers -> one scale; k-steps [0, KS) coordinate half, [KS, 2KS) geometry half
    {
        const float sc = pow2_scale(p->w_film[0], (int64_t)Hd * 2 * Hd, target);
        pack_acc(p->w_film[0], 2 * Hd, 0, Hd, Hd, 0, L.KS, sc, W_F0);
        pack_acc(p->w_film[0], 2 * Hd, Hd, Hd, Hd, L.KS, L.KS, sc, W_F0);
        invs[W_F0] = 1.f / sc;
    }
    for (int l = 1; l < 4; ++l) {
        const float sc = pow2_scale(p->w_film[l], (int64_t)Hd * Hd, target);
        pack_acc(p->w_film[l], Hd, 0, Hd, Hd, 0, L.KS, sc, W_F0 + l);
        invs[W_F0 + l] = 1.f / sc;
    }
    // colour layer: KS k-steps over the hidden features (columns 3..), one k-step over the view direction (columns
    // 0..2, natural order); one scale for the whole matrix (same accumulators), both inputs unscaled
    {
        const float sc = pow2_scale(p->w_color, (int64_t)Hd * (Hd + 3), target);
        pack_acc(p->w_color, Hd + 3, 3, Hd, Hd, 0, L.KS, sc, W_COLOR);
        x3t_pack_f16(p->w_color, Hd + 3, 0, 3, Hd, L.NT, tile_bytes_of(W_COLOR, L.KS, x2),
                     x2 ? x3t_kstep_off<true>(L.KS) : x3t_kstep_off<false>(L.KS), 1, sc, dst(W_COLOR), false);
        invs[W_COLOR] = 1.f / sc;
    }
    {
        const float sc = pow2_scale(p->w_feat, (int64_t)F * Hd, target);
        pack_acc(p->w_feat, Hd, 0, Hd, F, 0, L.KS, sc, W_FEAT);
        invs[W_FEAT] = 1.f / sc;
    }
    float* bias = reinterpret_cast<float*>(blob + L.bias);
    for (int nn = 0; nn < Hd; ++nn) {
        bias[ST_COORD * L.HdP + nn] = p->b_coord[nn];
        bias[ST_GEO * L.HdP + nn] = p->b_geo[nn];
        for (int l = 0; l < 4; ++l) bias[(ST_FILM0 + l) * L.HdP + nn] = p->b_film[l][nn];
        bias[ST_COLOR * L.HdP + nn] = p->b_color[nn];
    }
    float* bf = reinterpret_cast<float*>(blob + L.b_feat);
    for (int nn = 0; nn < F; ++nn) bf[nn] = p->b_feat[nn];
    // heads: one 32-row A tile whose rows 0..3 are sigma, r, g, b (own power-of-two scale each), K in accumulator order
    uint16_t* hw = reinterpret_cast<uint16_t*>(blob + L.head_w);
    float* hinv = reinterpret_cast<float*>(blob + L.head_inv);
    float* hb = reinterpret_cast<float*>(blob + L.head_b);
    for (int hd = 0; hd < 4; ++hd) {
        const float* w = hd == 0 ? p->w_sigma : p->w_rgb + (int64_t)(hd - 1) * Hd;
        const float sc = pow2_scale(w, Hd, target);
        hinv[hd] = 1.f / sc;
        hb[hd] = hd == 0 ? p->b_sigma[0] : p->b_rgb[hd - 1];
        if (x2) {      // rows 0..3 of the head tile as hi fragments + records (lane = 32 * hh + row)
            unsigned char* hbase = blob + L.head_w;
            for (int T = 0; T < L.KS / 2; ++T)
                for (int hh = 0; hh < 2; ++hh) {
                    float hi[16], lo[16];
                    const int lane = 32 * hh + hd;
                    for (int j = 0; j < 2; ++j)
                        for (int e = 0; e < 8; ++e) {
                            const int k = x3t_acc_k(2 * T + j, hh, e);
                            const float v = k < Hd ? w[k] * sc : 0.f;
                            const uint16_t h16 = x3t_f32_to_f16_rn(v);
                            hi[8 * j + e] = x3t_f16_to_f32(h16);
                            lo[8 * j + e] = v - hi[8 * j + e];
                            reinterpret_cast<uint16_t*>(hbase + (int64_t)(2 * T + j) * 2048)[lane * 8 + e] = h16;
                        }
                    unsigned rec[8];
                    x2_make_record(hi, lo, rec);
                    for (int j = 0; j < 2; ++j)
                        for (int d = 0; d < 4; ++d)
                            reinterpret_cast<unsigned*>(hbase + (int64_t)(2 * T + j) * 2048 + 1024)[lane * 4 + d] = rec[4 * j + d];
                }
            continue;
        }
        for (int ks = 0; ks < L.KS; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                for (int e = 0; e < 8; ++e) {
                    const int k = x3t_acc_k(ks, hh, e);
                    const float v = k < Hd ? w[k] * sc : 0.f;
                    const uint16_t hi = x3t_f32_to_f16_rn(v), lo = x3t_f32_to_f16_rn(v - x3t_f16_to_f32(hi));
                    const int64_t base = ((int64_t)ks * 2) * 64 * 8 + (32 * hh + hd) * 8 + e;      // lane = 32*hh + row
                    hw[base] = hi;
                    hw[base + 64 * 8] = lo;
                }
        hinv[hd] = 1.f / sc;
        hb[hd] = hd == 0 ? p->b_sigma[0] : p->b_rgb[hd - 1];
    }
    return H3D_OK;
}
}  // namespace

extern "C" int h3d_neural_field_x3t_tier(const void* packed, const float* points, const float* geo, const float* dirs,
                                         const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                                         int geo_stride, float input_scaler, int products, h3d_stream_t stream) {
    int rc = check_x3t(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(products == 1 || products == 3 || products == 4,
                "h3d_neural_field_x3t_tier: products must be 1 (plain f16), 3 (split f16) or 4 (x2: blob from h3d_field_pack_x2t)");
    H3D_REQUIRE(out, "h3d_neural_field_x3t: null output");
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const unsigned char*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase; A.out = out;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = 64; A.input_scaler = input_scaler;
    A.L = make_layout(Hd, F);
    const int64_t groups = (N + 63) / 64;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_neural_field_x3t: N too large");
    return products == 3 ? launch<false, 3>(A, B, groups, static_cast<hipStream_t>(stream))
         : products == 4 ? launch<false, 4>(A, B, groups, static_cast<hipStream_t>(stream))
                         : launch<false, 1>(A, B, groups, static_cast<hipStream_t>(stream));
}

extern "C" int h3d_neural_field_x3t(const void* packed, const float* points, const float* geo, const float* dirs,
                                    const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                                    int geo_stride, float input_scaler, h3d_stream_t stream) {
    return h3d_neural_field_x3t_tier(packed, points, geo, dirs, freq, phase, out, B, N, Hd, F, geo_stride, input_scaler, 3, stream);
}

extern "C" int h3d_render_fused_x3t_tier(const void* packed, const float* points, const float* geo, const float* dirs,
                                         const float* freq, const float* phase, const float* z_vals, const float* noise,
                                         float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                                         int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                                         int products, h3d_stream_t stream) {
    const int64_t N = (int64_t)R * S;
    int rc = check_x3t(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(products == 1 || products == 3 || products == 4,
                "h3d_render_fused_x3t_tier: products must be 1 (plain f16), 3 (split f16) or 4 (x2: blob from h3d_field_pack_x2t)");
    H3D_REQUIRE(z_vals && feats && depth && weights, "h3d_render_fused_x3t: null pointer");
    H3D_REQUIRE(clamp_mode == 0 || clamp_mode == 1, "h3d_render_fused_x3t: clamp_mode must be 0 (relu) or 1 (softplus)");
    H3D_REQUIRE(R >= 0 && S >= 1, "h3d_render_fused_x3t: bad R=%d S=%d", R, S);
    const bool ok_s = (S >= 8 && S <= 64 && (S & (S - 1)) == 0) || (S > 64 && S % 64 == 0);
    if (!ok_s) {
        h3d::set_error("h3d_render_fused_x3t: S=%d unsupported by the fused kernel (needs 8,16,32,64 or a multiple of 64); "
                       "use h3d_neural_field_x3t + h3d_ray_integrate", S);
        return H3D_EUNSUPPORTED;
    }
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const unsigned char*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase;
    A.z_vals = z_vals; A.noise = noise; A.feats = feats; A.depth = depth; A.weights = weights;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = S; A.input_scaler = input_scaler;
    A.clamp_mode = clamp_mode; A.last_back = last_back; A.white_back = white_back;
    A.L = make_layout(Hd, F);
    const int group = S > 64 ? S : 64;
    const int64_t groups = (N + group - 1) / group;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_render_fused_x3t: too many rays");
#ifdef H3D_EXPERIMENT_TRACE
    {   // development build: dump the cycle trace of workgroup (1000, 3) to $H3D_TRACE_FILE after every launch
        static unsigned long long* tb = nullptr;
        if (!tb) (void)hipMalloc(&tb, 4096 * 8);
        (void)hipMemset(tb, 0, 4096 * 8);
        A.out = reinterpret_cast<float*>(tb);
        const int rc2 = products == 3 ? launch<true, 3>(A, B, groups, static_cast<hipStream_t>(stream))
                      : products == 4 ? launch<true, 4>(A, B, groups, static_cast<hipStream_t>(stream))
                                      : launch<true, 1>(A, B, groups, static_cast<hipStream_t>(stream));
        (void)hipDeviceSynchronize();
        static unsigned long long host[4096];
        (void)hipMemcpy(host, tb, sizeof(host), hipMemcpyDeviceToHost);
        if (const char* f = getenv("H3D_TRACE_FILE")) {
            if (FILE* fp = fopen(f, "w")) {
                unsigned long long t0 = host[0] >> 8, prev = t0;
                for (int i = 0; i < 4096 && host[i]; ++i) {
                    const unsigned long long tt = host[i] >> 8;
                    fprintf(fp, "%llu %llu +%llu\n", host[i] & 255ull, tt - t0, tt - prev);
                    prev = tt;
                }
                fclose(fp);
            }
        }
        return rc2;
    }
#endif
    return products == 3 ? launch<true, 3>(A, B, groups, static_cast<hipStream_t>(stream))
         : products == 4 ? launch<true, 4>(A, B, groups, static_cast<hipStream_t>(stream))
                         : launch<true, 1>(A, B, groups, static_cast<hipStream_t>(stream));
}

extern "C" int h3d_render_fused_x3t(const void* packed, const float* points, const float* geo, const float* dirs,
                                    const float* freq, const float* phase, const float* z_vals, const float* noise,
                                    float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                                    int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                                    h3d_stream_t stream) {
    return h3d_render_fused_x3t_tier(packed, points, geo, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S,
                                     Hd, F, geo_stride, input_scaler, clamp_mode, last_back, white_back, 3, stream);
}
