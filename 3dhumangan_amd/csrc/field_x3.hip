// A5 (+A6 fused): FiLM-SIREN field on the f16 matrix cores with split ("x3") operands, for gfx950.
//
// Reference semantics as neural_field.hip (lib/implicit_funcitions/modulated.py:41-75,
// lib/generators/volume_rendering.py:12-56).  Same math, different engine:
//
//   precision   every fp32 operand is split x*s = hi + lo with hi = f16(x*s), lo = f16(x*s - hi) (s a power of
//               two that keeps both halves in the normal f16 range) and a product is evaluated as
//               hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 22 significant bits per
//               operand, i.e. fp32-class results at 16/3 of the fp32 MFMA rate.  A single f16 product misses the
//               1e-3 budget (5e-3 after the freq~45 sines), which is why the split is needed.
//   dataflow    a wavefront owns 32 samples and ALL output features.  Weights are the MFMA A operand (rows =
//               features), activations the B operand (columns = samples), so the accumulator of a lane holds one
//               sample's features and one v_permlane32_swap per register pair turns it into the next layer's B
//               fragments: activations never touch LDS or HBM between layers.  Two accumulator sets (2 x 128
//               AGPRs) ping-pong through the layers, so the epilogue of layer l (FiLM sine + split) runs tile by
//               tile INSIDE the GEMM of layer l+1 -- k-steps 2t, 2t+1 only need tile t -- hidden behind its MFMAs
//               (x3_common.hpp: gemm_x3_roll hook).  Every input enters as a GEMM: the coordinate layer (K = 3),
//               the geometry-feature layer (K = 31) and the view direction (one extra k-step of the colour layer).
//               The density and colour heads ride along as a ninth 32-row tile of the colour / feature GEMMs.
//   weights     one linear stream in consumption order (2 KB per tile and k-step, MFMA A-fragment order); the four
//               waves of a workgroup pull it ONCE from L2 into an LDS ring with global_load_lds (LDS-DMA), several
//               k-steps ahead and straight through the epilogues; one raw s_barrier per k-step.
//   sine        the FiLM affine and the 1/2pi of v_sin_f32 (argument in revolutions) are folded into two per-channel
//               table values, u = acc*A1 + A0, y = v_sin(u): 1 FMA + 1 transcendental per activation.  fl(u) carries
//               the same |y| * 2^-24 argument error as the reference's own fp32 evaluation of f*(Wx+b)+p.
//   fused A6    the feature head is evaluated with the operands swapped (rows = samples), so the weighted sum over
//               the samples of a ray is a sum over accumulator registers; compositing weights come from a
//               segmented 32-lane product scan (S = 8..32: 32/S rays per wave step; S = 64, 128: carry).
//
// X2 variant (template parameter, h3d_neural_field_x2 / h3d_render_fused_x2): the hidden-layer contractions (FiLM 0-3, the
// hidden part of the colour layer, the feature head) run in the "x2" arithmetic of x3_common.hpp -- one f16 product + one
// block-scaled fp6 product for the two cross terms -- which halves the matrix-pipe time and the energy per contraction; the
// input layers (K = 3 / 31 / view direction) and the 1-row heads stay on three f16 products.  Error model and budget:
// tests/x2_emulation.py, tests/test_x2_error_model_cpu.py (2e-5 at the operator boundary at width 256).
#include "x3_common.hpp"
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

using namespace h3d;

namespace {

typedef F16::vec8 half8;
typedef F16::vec2 half2v;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kSA = 1.f;         // activation scale: none -- the matrix cores honour f16 subnormals (tools/probes/denorm_probe.hip),
                                   // so hi = f16(y), lo = f16(y - hi) keeps |error| <= 2^-25 for |y| <= 1 without pre-scaling
constexpr float kSIn = 64.f;       // input scale (coords / geometry features, |x| < 1000)
#ifndef H3D_FIELD_LOOK
#define H3D_FIELD_LOOK 2
#endif
#ifndef H3D_FIELD_VALU
#define H3D_FIELD_VALU 4
#endif
constexpr int kLookF = H3D_FIELD_LOOK;          // weight-fragment look-ahead in tile pairs
constexpr int kValuF2 = 0;                      // x2: no forced VALU / MFMA interleave (measured: 17.19 -> 16.86 ms vs 4; 3 and 8 in between)
constexpr int kValuF = H3D_FIELD_VALU;          // VALU instructions slotted behind each MFMA of a section carrying epilogue work

// per-step activation tables (A1, A0) in LDS
enum { ST_COORD = 0, ST_GEO, ST_FILM0, ST_FILM1, ST_FILM2, ST_FILM3, ST_COLOR, ST_COUNT };
// weight matrices in STREAM (consumption) order
enum { W_COORD = 0, W_F0A, W_GEO, W_F0B, W_F1, W_F2, W_F3, W_COLOR, W_FEAT, W_COUNT };

struct LayoutX3 {        // offsets in BYTES into the blob (all multiples of 16)
    int HdP, FP, NT, KS, stages, head_planes;
    int64_t w[W_COUNT];
    int64_t inv_scale;   // float[W_COUNT]: 1 / (weight scale * input scale) per matrix
    int64_t bias;        // float[ST_COUNT][HdP]
    int64_t b_feat;      // float[FP]
    int64_t head_w;      // f16 [4 heads: sigma, r, g, b][planes: hi, lo (, hi * 2^-12: x2)][KS][2 halves][8]  (A-fragment rows of the head tile)
    int64_t head_inv;    // float[4] 1/(weight scale * kSA)
    int64_t head_b;      // float[4]
    int64_t total;
};

int tiles_for(int Hd) { int nt = 4; while (nt * 32 < Hd) nt *= 2; return nt; }   // engine is built for 4 or 8 tiles

int stages_of(int wi, int KS) { return wi == W_COORD ? 1 : wi == W_GEO ? 2 : wi == W_COLOR ? KS + 1 : KS; }

LayoutX3 make_layout(int Hd, int F, bool x2 = false) {
    LayoutX3 L;
    L.head_planes = x2 ? 3 : 2;
    const int w = Hd > F ? Hd : F;
    L.NT = tiles_for(w);
    L.HdP = L.FP = L.NT * 32;
    L.KS = L.HdP / 16;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 15) / 16 * 16; return r; };
    const int64_t per_ks = (int64_t)L.NT * 2 * 64 * 16;
    L.stages = 0;
    for (int i = 0; i < W_COUNT; ++i) { L.w[i] = take(stages_of(i, L.KS) * per_ks); L.stages += stages_of(i, L.KS); }
    L.inv_scale = take(4 * W_COUNT);
    L.bias = take(4 * (int64_t)ST_COUNT * L.HdP);
    L.b_feat = take(4 * (int64_t)L.FP);
    L.head_w = take(2 * (int64_t)4 * L.head_planes * L.KS * 16);
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
    int R, log2S;        // fused: rays per batch item; log2(S) when S <= 32 (a power of two), else -1
    int n_groups;        // unit groups (4 wave units each) per sample; a workgroup walks blockIdx.x, + gridDim.x, ..
    float input_scaler;
    LayoutX3 L;
    // GEOIN (A4 inside the kernel): the nearest-vertex index of every sample and the pose tables the features are built from
    const int* nn_index;         // [B, N]
    const float* joints;         // [B, 24, 3]
    const float* vertices;       // [B, V, 3]
    const float* tpose;          // [B, V, 3]
    const float* vertex_ik;      // [B, V, 16] blended inverse bone transforms
    int V, legacy_mode;
    // Refinement of ill-conditioned last samples (round 6, fused kernels): the reference gives a ray's last sample delta = 1e9
    // (lib/generators/volume_rendering.py:21), so its alpha is 0 or 1 by the SIGN of its density -- a density within the
    // arithmetic's error of zero flips the ray's background term.  ref_mode 1 (the x2 launch): a wave unit (one ray when S > 32)
    // whose ray has |sigma_last| <= ref_eps * max(max_s |sigma_s|, ref_scale) appends its index to ref_list[b][..] (ref_count[b]
    // counts them, also beyond ref_cap).  ref_mode 2 (the x3 launch behind it): the launch covers exactly the listed units.
    int* ref_list;               // [B, ref_cap] wave-unit indices
    int* ref_count;              // [B]
    int ref_cap, ref_mode;
    float ref_eps, ref_scale;
};

constexpr int kHeadPad = 64;       // bytes behind every head's fragment rows in LDS (bank staggering)
constexpr int kJointRows = 40;     // GEOIN: LDS table of 3 + 24 + 13 float4 rows (joints at rows 3..26, zeros around them)

// the oracle's squared distance, (dx*dx + dy*dy) + dz*dz without contraction (geo_features.hip: sqdist_exact)
__device__ __forceinline__ float sqdist_exact(float px, float py, float pz, float vx, float vy, float vz) {
    const float dx = __fsub_rn(px, vx), dy = __fsub_rn(py, vy), dz = __fsub_rn(pz, vz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ float density(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? x : log1pf(expf(x));
    return fmaxf(x, 0.f);
}

// two fp32 activations -> packed f16 hi halves (returned) and packed f16 lo halves, plain (non-packed) VALU only:
// VOP3P instructions (v_pk_*, v_fma_mix) beside MFMAs cost several times their issue slot on this chip, and hipcc's SLP
// vectoriser would turn the two subtractions into one v_pk_add_f32, hence the asm.
__device__ __forceinline__ unsigned split2_act(float a, float b, unsigned& lo) {
    const half2v h2 = __builtin_convertvector(f32x2{a, b}, half2v);
    const float fa = (float)h2.x, fb = (float)h2.y;
    float la, lb;
    asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(fa));
    asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(fb));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{la, lb}, half2v));
    return __builtin_bit_cast(unsigned, h2);
}

// x2: the lo halves travel multiplied by rho = 2^12 (they only feed the fp6 conversion and the heads' scaled plane)
__device__ __forceinline__ unsigned split2_act_x2(float a, float b, unsigned& lo) { return split2_x2_bounded(a, b, lo); }   // |sin| <= 1

// two fp32 (already scaled) -> packed f16 hi halves (returned) and packed f16 lo halves
__device__ __forceinline__ unsigned split2_f16(float a, float b, unsigned& lo) {
    const half2v h2 = __builtin_convertvector(f32x2{a, b}, half2v);
    const float fa = (float)h2.x, fb = (float)h2.y;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a - fa, b - fb}, half2v));
    return __builtin_bit_cast(unsigned, h2);
}

// 8 fp32 values of one lane -> hi / lo B-fragments
__device__ __forceinline__ void split8(const float (&v)[8], float scale, half8& fh, half8& fl) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) hw[e / 2] = split2_f16(v[e] * scale, v[e + 1] * scale, lw[e / 2]);
    fh = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    fl = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// K order of every GEMM that consumes accumulators ("acc order"): the accumulator registers a lane holds for tile t ARE
// its B-fragment elements -- register r = 8j + e of tile t is element e of k-step 2t + j -- so no cross-lane relayout is
// needed; the host packs the weights' K dimension accordingly (k_of below / acc_k in the packer):
//     feature of k-slot (h, e) of k-step ks:  32*(ks/2) + (e & 3) + 8*(2*(ks & 1) + (e >> 2)) + 4*h
__device__ __forceinline__ void set_word(half8& f, int w, unsigned v) {
    u32x4 t = __builtin_bit_cast(u32x4, f);
    t[w] = v;
    f = __builtin_bit_cast(half8, t);
}

// Producer of the next layer's B fragments from a feature-major accumulator set: y = v_sin(acc * A1[n] + A0[n])
// (FiLM affine, bias, de-scaling and 1/2pi folded into the two LDS tables), split to f16 hi/lo.  The work of one
// 32-feature tile is cut into eight chunks of two activations so that the consuming GEMM can hide one chunk behind
// each of the eight tile-pair sections that precede the tile's first use; table values are fetched one chunk ahead.
template <int NT, bool X2 = false>
struct FilmProducer {
    f32x16 (&src)[NT];
    half8 (&xh)[2 * NT + 1];
    half8 (&xl)[2 * NT + 1];
    i32x8 (&b6)[NT];          // x2: the fp6 activation record of each K-tile
    lds_ptr tab;              // lane base (+ 32 h bytes: lane_base) of the LDS table [HdP/2][4]: A1[n], A1[n+1], A0[n], A0[n+1]
    f32x4 tv;
    float sv[8];              // half a tile of the source accumulators, read from the AGPRs in one batch: a
                              // v_accvgpr_read issued between MFMAs waits for the matrix pipe, so 2 batches per tile
                              // instead of 8 pairs

    template <int TILE, int C>
    static constexpr int chan() { return TILE * 32 + (C / 2) * 8 + (C % 2) * 2; }      // + 4 h: in the lane base
    __device__ __forceinline__ void prime() { tv = ldt4(tab, 2 * chan<0, 0>()); }
    // The table row of chunk C + 1, read at the TOP of the section that runs chunk C (gemm_x2_roll's PRE hook), in front of the
    // section's burst of weight-fragment reads: one section later the wait for it leaves that burst in flight.  (Read inside
    // chunk C -- after the burst in program order, or sunk there by instruction selection -- it is the YOUNGEST LDS read at its
    // use and its wait is s_waitcnt lgkmcnt(0): the look-ahead drained once per section; round 5, seen in the ISA.)
    f32x4 tn;
    template <int TILE, int C>
    __device__ __forceinline__ void prefetch() {
        if constexpr (C < 7) tn = ldt4_pinned(tab, 2 * chan<TILE, C + 1>());
        else if constexpr (TILE + 1 < NT) tn = ldt4_pinned(tab, 2 * chan<TILE + 1, 0>());
    }
    template <int TILE, int C, bool PRE = false>
    __device__ __forceinline__ void chunk() {
        if constexpr (C == 0) pin1(src[TILE]);
        constexpr int rg = C / 2;
        if constexpr (C % 4 == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[i] = src[TILE][(C / 4) * 8 + i];
        }
        const float s0 = sv[(C % 4) * 2], s1 = sv[(C % 4) * 2 + 1];
        const float u0 = fmaf(s0, tv.x, tv.z);
        const float u1 = fmaf(s1, tv.y, tv.w);
        if constexpr (PRE) {
            tv = tn;                                        // fetched by prefetch<TILE, C>() at the top of this section
        } else {
            if constexpr (C < 7) tv = ldt4(tab, 2 * chan<TILE, C + 1>());
            else if constexpr (TILE + 1 < NT) tv = ldt4(tab, 2 * chan<TILE + 1, 0>());
        }
        unsigned lo;
        const unsigned hi = X2 ? split2_act_x2(__builtin_amdgcn_sinf(u0), __builtin_amdgcn_sinf(u1), lo)
                               : split2_act(__builtin_amdgcn_sinf(u0), __builtin_amdgcn_sinf(u1), lo);
        // registers 4*rg + 2*(C%2) + {0, 1} of the tile = k-step 2*TILE + (rg >> 1), word 2*(rg & 1) + C%2
        set_word(xh[2 * TILE + (rg >> 1)], 2 * (rg & 1) + (C % 2), hi);
        set_word(xl[2 * TILE + (rg >> 1)], 2 * (rg & 1) + (C % 2), lo);
    }
    // x2: fp6 record of K-tile TILE from its four finished f16 fragments (runs in the first section of k-step 2 * TILE of the
    // consuming GEMM; the record is first used one k-step later)
    template <int TILE>
    __device__ __forceinline__ void convert() {
        if constexpr (X2) b6[TILE] = x2_record(xl[2 * TILE], xl[2 * TILE + 1], xh[2 * TILE], xh[2 * TILE + 1]);
    }
};

struct NoProducer {
    __device__ __forceinline__ void prime() {}
    template <int TILE, int C, bool PRE = false> __device__ __forceinline__ void chunk() {}
    template <int TILE, int C> __device__ __forceinline__ void prefetch() {}
    template <int TILE> __device__ __forceinline__ void convert() {}
};

// Tile-0 work of the NEXT layer's producer (its source is this GEMM's destination): tiles 0 and 1 of dst are final after
// the first tile-pair section of the last k-step, so the remaining P-1 sections of that k-step carry the eight chunks
// -- no layer starts with an exposed epilogue.
template <int P, int S_LAST, int G, typename NEXT>
__device__ __forceinline__ void next_tile0(NEXT& next) {
    constexpr int s = G / P, p = G % P;
    if constexpr (s == S_LAST) {
        if constexpr (p == 0) next.prime();
        constexpr int per = (8 + P - 2) / (P - 1);
        if constexpr (p >= 1) {
            static_for<0, per>([&](auto q) __attribute__((always_inline)) {
                constexpr int c = (p - 1) * per + decltype(q)::value;
                if constexpr (c < 8) next.template chunk<0, c>();
            });
        }
    }
}

// One layer: dst (+)= W * frags(producer(src)).  Tile 0 of the producer has already run inside the previous GEMM
// (next_tile0), tile t+1 runs inside the sections of k-steps 2t, 2t+1, and tile 0 of `next` inside the last k-step.
// HEAD: the four head rows (sigma, r, g, b) ride along as a ninth tile accumulated in hacc (feature-major: row = head,
// column = sample), A fragments from LDS, one k-step of look-ahead.
template <int NT, int KSG, bool ZERO, bool SWAP, bool HEAD, bool X2, typename RING, typename PROD, typename NEXT>
__device__ __forceinline__ void layer(f32x16 (&dst)[NT], half8 (&xh)[2 * NT + 1], half8 (&xl)[2 * NT + 1], i32x8 (&b6)[NT], RING& ring,
                                      PROD& prod, NEXT& next, f32x16& hacc, const unsigned char* head_lds, int lane) {
    constexpr int KS = 2 * NT, P = NT / 2, W = NT, PER = 8 / W, PL = X2 ? 3 : 2;
    u32x4 hwh, hwl, hws;
    // head A fragment of k-step s: lane (row m, half h) reads head (m & 3), plane hi / lo (/ hi * 2^-12: the x2 lo
    // fragments are pre-multiplied by 2^12): [head][plane][KS][half][16 B]
    const lds_ptr hbase = lane_base(head_lds, (lane & 3) * (PL * KS * 32 + kHeadPad) + (lane >> 5) * 16);
    auto load_head = [&](int s) __attribute__((always_inline)) {
        hwh = lds_ld<u32x4>(hbase + s * 32);
        hwl = lds_ld<u32x4>(hbase + (KS + s) * 32);
        if constexpr (X2) hws = lds_ld<u32x4>(hbase + (2 * KS + s) * 32);
    };
    if constexpr (HEAD) load_head(0);
    __builtin_amdgcn_sched_barrier(0);
#ifdef H3D_FIELD_NO_PREFETCH_PIN
    constexpr bool kPre = false;
#else
    constexpr bool kPre = X2 && PER == 1 && !std::is_same<PROD, NoProducer>::value;       // one chunk per section: one prefetch per section
#endif
    auto pre = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int t = g / W + 1, j = g % W;
        if constexpr (kPre && t < NT) prod.template prefetch<t, j * PER>();
    };
    auto hook = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int t = g / W + 1, j = g % W;
        if constexpr (X2 && j == 0 && t - 1 < NT) prod.template convert<t - 1>();      // first section of k-step 2 * (t - 1)
        if constexpr (t < NT) {
            static_for<0, PER>([&](auto q) __attribute__((always_inline)) {
                prod.template chunk<t, j * PER + decltype(q)::value, kPre>();
            });
        }
        next_tile0<P, KSG - 1, g>(next);
        if constexpr (HEAD) {
            constexpr int s = g / P, p = g % P;
            if constexpr (p == 0 && s < KS) {
                const half8 ah = __builtin_bit_cast(half8, hwh), al = __builtin_bit_cast(half8, hwl);
                if constexpr (s == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    hacc = F16::mfma(ah, xh[s], zero);
                } else {
                    hacc = F16::mfma(ah, xh[s], hacc);
                }
                hacc = F16::mfma(X2 ? __builtin_bit_cast(half8, hws) : ah, xl[s], hacc);
                hacc = F16::mfma(al, xh[s], hacc);
            }
            if constexpr (p == 1 % P && s + 1 < KS) load_head(s + 1);
        }
    };
    if constexpr (X2) {
        constexpr int KS3 = KSG - KS;
        const half8 tail[1] = {xl[KS]};                 // view direction k-step: assembled from memory, lo unscaled
        gemm_x2_roll<NT, KS, KS3, 2 * NT + 1, NT, SWAP, kLookF, kValuF2, ZERO>(dst, xh, b6, tail, ring, hook, pre);
    } else {
        gemm_x3_roll<F16, NT, KSG, 2 * NT + 1, SWAP, kLookF, kValuF, ZERO>(dst, xh, xl, ring, hook);
    }
}

// Input GEMM (1 or 2 k-steps, fragments prebuilt) with the tile-0 work of the layer that consumes its result.
// (its own fragment arrays: the consumer's tile-0 chunks write xh[0], xh[1] while this GEMM is still running)
template <int NT, int KSG, typename RING, typename NEXT>
__device__ __forceinline__ void input_layer(f32x16 (&dst)[NT], const half8 (&ih)[2], const half8 (&il)[2], RING& ring, NEXT& next) {
    gemm_x3_roll<F16, NT, KSG, 2, false, kLookF, kValuF, true>(dst, ih, il, ring, [&](auto gc) __attribute__((always_inline)) {
        next_tile0<NT / 2, KSG - 1, decltype(gc)::value>(next);
    });
}


#ifndef H3D_FIELD_RINGX2
#define H3D_FIELD_RINGX2 8
#endif
constexpr int kRingX2 = H3D_FIELD_RINGX2;       // x2: one more buffer, the refill lags one stage (WeightRing LAG = 1)

template <int NT, bool FUSED, bool X2, bool GEOIN = false>
__global__ __launch_bounds__(256, 1) void field_x3_kernel(Args A) {
    constexpr int KS = 2 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LayoutX3& L = A.L;
    const int HdP = L.HdP;
    float* tab0 = smem;                                 // [ST_COUNT][2][HdP]  A1 / A0 per step
    float* tfeat0 = tab0 + ST_COUNT * 2 * HdP;          // [HdP] feature-head bias
    float* scratch = tfeat0 + HdP;                      // [4 waves][64]: compositing weights, background terms
    float* joint0 = scratch + 4 * 64;                   // GEOIN: [kJointRows] float4 (x, y, z, 0): joints at rows 3 .. 26
    unsigned char* head0 = reinterpret_cast<unsigned char*>(joint0 + (GEOIN ? kJointRows * 4 : 0));      // head A-fragment rows
    unsigned char* ring_lds = head0 + 4 * ((X2 ? 3 : 2) * KS * 32 + kHeadPad);           // [ring depth][NT*2 KB]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    // refinement launch (ref_mode 2): the units the x2 launch listed for this batch item; usually none -- leave before the tables
    int ref_units = 0;
    if constexpr (FUSED) {
        if (A.ref_mode == 2) {
            ref_units = __builtin_amdgcn_readfirstlane(min(A.ref_count[b], A.ref_cap));
            if ((int)blockIdx.x * 4 >= ref_units) return;
        }
    }
    (void)ref_units;
    const int Hd = A.Hd, F = A.F, S = A.S;
    const int64_t N = A.N;
    const float* __restrict__ invs = reinterpret_cast<const float*>(A.blob + L.inv_scale);
    // ---- per-sample-of-the-batch activation tables in LDS, once per workgroup:
    //      y = sin(f * (acc*inv + bias) + p) = v_sin(acc * A1 + A0),  A1 = inv*f/2pi,  A0 = (bias*f + p)/2pi
    {
        const unsigned char* __restrict__ blob = A.blob;
        const float* __restrict__ bias = reinterpret_cast<const float*>(blob + L.bias);
        const float* __restrict__ fr = A.freq + (int64_t)b * 4 * Hd;
        const float* __restrict__ ph = A.phase + (int64_t)b * 4 * Hd;
        const float* __restrict__ bf = reinterpret_cast<const float*>(blob + L.b_feat);
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
                const int wi = st == ST_COORD ? W_COORD : st == ST_GEO ? W_GEO : st == ST_FILM0 ? W_F0A
                             : st == ST_COLOR ? W_COLOR : W_F1 + (st - ST_FILM1);
                a1 = invs[wi] * ff * inv2pi;
                a0 = fmaf(bias[st * HdP + n], ff, pp) * inv2pi;
            }
            float* q = tab0 + st * 2 * HdP + (n >> 1) * 4 + (n & 1);
            q[0] = a1;
            q[2] = a0;
        }
        for (int idx = t; idx < HdP; idx += 256) tfeat0[idx] = idx < F ? bf[idx] : 0.f;
        if constexpr (GEOIN) {
            if (t < kJointRows * 4) {
                const int row = t >> 2, c = t & 3, j = row - 3;
                joint0[t] = (j >= 0 && j < 24 && c < 3) ? A.joints[((int64_t)b * 24 + j) * 3 + c] : 0.f;
            }
        }
        const u32x4* hsrc = reinterpret_cast<const u32x4*>(blob + L.head_w);
        u32x4* hdst = reinterpret_cast<u32x4*>(head0);
        // 64 bytes of padding behind every head: the four heads' rows would otherwise start 1536 (3072) bytes apart -- the same
        // banks -- and every head-fragment read (lanes of a 16-lane group address all four heads) was a 4-way conflict
        constexpr int per_head = (X2 ? 3 : 2) * KS * 2;               // 16-byte units per head
        for (int idx = t; idx < 4 * per_head; idx += 256) hdst[idx + (idx / per_head) * (kHeadPad / 16)] = hsrc[idx];
    }
    __syncthreads();

    const float* __restrict__ head_inv = reinterpret_cast<const float*>(A.blob + L.head_inv);
    const float* __restrict__ head_b = reinterpret_cast<const float*>(A.blob + L.head_b);
    float inv_f = invs[W_FEAT];
    float hi0 = head_inv[0], hi1 = head_inv[1], hi2 = head_inv[2], hi3 = head_inv[3];
    float hb0 = head_b[0], hb1 = head_b[1], hb2 = head_b[2], hb3 = head_b[3];
    // uniform constants that only ever feed per-lane arithmetic: kept in vector registers (see the pointers below)
    asm volatile("" : "+v"(inv_f), "+v"(hi0), "+v"(hi1), "+v"(hi2), "+v"(hi3), "+v"(hb0), "+v"(hb1), "+v"(hb2), "+v"(hb3));
    float* wl_lds = scratch + wave * 64;      // [32] weights, [32] background

    const int unit = FUSED ? (S > 32 ? S : 32) : 32;          // samples a wave walks per unit (whole rays when fused)
    const int steps = unit / 32;
    const int seglen = FUSED ? (S < 32 ? S : 32) : 32;
    H3D_TRACE_INIT();
    H3D_TRACE(0);
    typedef typename std::conditional<X2, WeightRing<NT, kRingX2, 1>, WeightRing<NT>>::type Ring;
    Ring ring;
    ring.init(A.blob + L.w[0], ring_lds, L.stages, wave, lane);
    int n_groups = A.n_groups;
    if constexpr (FUSED) {
        if (A.ref_mode == 2) n_groups = (ref_units + 3) >> 2;
    }
    asm volatile("" : "+s"(n_groups));           // pinned: not re-loaded from the kernarg segment inside the loop
    // refinement state in vector registers (see the pointers below): mode, threshold factor, floor of the scale, list of this item
    int ref_mode = FUSED ? A.ref_mode : 0;
    float ref_eps = A.ref_eps, ref_scale = A.ref_scale;
    typedef __attribute__((address_space(1))) int* gi32;
    gi32 ref_list = (gi32)A.ref_list + (int64_t)b * A.ref_cap, ref_count = (gi32)A.ref_count + b;
    int ref_cap = A.ref_cap;
    asm volatile("" : "+v"(ref_mode), "+v"(ref_eps), "+v"(ref_scale), "+v"(ref_list), "+v"(ref_count), "+v"(ref_cap));
    // The per-step global pointers live in VECTOR registers (the kernel has ~60 to spare, and every one of them is only ever
    // used in per-lane address arithmetic): as scalar values they were the bulk of the kernel's 70-85 spilled SGPRs, each
    // reloaded with v_readlane where it was used.
    // (address-space-1 pointer types: a laundered generic pointer would turn every access into a FLAT one)
    typedef const __attribute__((address_space(1))) float* gcf;
    typedef __attribute__((address_space(1))) float* gf;
    typedef const __attribute__((address_space(1))) int* gci;
    gcf a_points = (gcf)A.points, a_geo = (gcf)A.geo, a_dirs = (gcf)A.dirs, a_z = (gcf)A.z_vals, a_noise = (gcf)A.noise;
    gf a_feats = (gf)A.feats, a_depth = (gf)A.depth, a_weights = (gf)A.weights, a_out = (gf)A.out;
    gci a_nn = (gci)A.nn_index;
    gcf a_vik = (gcf)A.vertex_ik, a_tpose = (gcf)A.tpose, a_verts = (gcf)A.vertices;
    asm volatile("" : "+v"(a_points), "+v"(a_geo), "+v"(a_dirs), "+v"(a_z), "+v"(a_noise), "+v"(a_feats), "+v"(a_depth),
                      "+v"(a_weights), "+v"(a_out));
    if constexpr (GEOIN) asm volatile("" : "+v"(a_nn), "+v"(a_vik), "+v"(a_tpose), "+v"(a_verts));

    // Persistent workgroups: the sample's tables above are built once, then the workgroup walks the unit groups blockIdx.x,
    // blockIdx.x + gridDim.x, .. with the weight ring running across them (the stream wraps at the end of every step).
#pragma unroll 1
    for (int ug = blockIdx.x; ug < n_groups; ug += gridDim.x) {
    int64_t uidx = (int64_t)ug * 4 + wave;                  // this wave's unit
    if constexpr (FUSED) {
        if (ref_mode == 2) {                                // refinement launch: the unit listed in slot ug * 4 + wave (or none)
            const int slot = ug * 4 + wave;
            uidx = slot < ref_units ? (int64_t)__builtin_amdgcn_readfirstlane(ref_list[slot < ref_units ? slot : 0]) : N;      // N * unit >= N: idle
        }
    }
    const int64_t u0 = uidx * unit;                         // may lie beyond N: such waves only keep the ring going
    // ray of sample n (fused): no 64-bit division -- S > 32: one ray per wave unit; S <= 32: S is a power of two
    const int64_t unit_ray = (int64_t)b * A.R + uidx;
    float smax = 0.f;                                       // largest |density| of the unit's rays so far (ref_mode 1)
    auto ray_of = [&](int64_t nn) -> int64_t { return A.log2S < 0 ? unit_ray : (int64_t)b * A.R + (nn >> A.log2S); };

    float carryT = 1.f, carryW = 0.f, carryD = 0.f, rgbacc = 0.f;
    float rayacc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) rayacc[i] = 0.f;

    f32x16 X[NT], Y[NT];          // the two accumulator sets

    for (int si = 0; si < steps; ++si) {
        const int64_t n0 = u0 + (int64_t)si * 32;
        const int64_t n = n0 + m;
        const bool ok = n < N;
        const int64_t gi = (int64_t)b * N + (ok ? n : N - 1);      // clamped: out-of-range lanes load valid memory
        const bool last_step = si == steps - 1;
        // The LDS tables do not depend on the step: launder an opaque zero offset so the compiler does not hoist
        // hundreds of loop-invariant loads out of the step loop (LICM) and spill them.
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* tab = tab0 + opaque;
        const float* tfeat = tfeat0 + opaque;
        const unsigned char* head_lds = head0 + opaque;

        H3D_TRACE(6);
        half8 xh[KS + 1], xl[KS + 1];
        i32x8 b6[NT];
        f32x16 hacc;
        NoProducer none;

        // ---- inputs as B fragments: coordinates (K = 3), geometry features (K = 31), view direction (K = 3)
        half8 ch, cl, gh[2], gl[2];
        {
            gcf p = a_points + gi * 3;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < 3 && h == 0 && ok) ? p[e < 3 ? e : 0] * A.input_scaler : 0.f;
            split8(v, kSIn, ch, cl);
            if constexpr (!GEOIN) {
                gcf g = a_geo + gi * A.geo_stride;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = ks * 16 + h * 8 + e;
                        v[e] = (k < 31 && ok) ? g[k < 31 ? k : 0] : 0.f;
                    }
                    split8(v, kSIn, gh[ks], gl[ks]);
                }
            } else {
                // ---- A4 in place (lib/components/smpl.py:210-249; the arithmetic of geo_features.hip's tail): this lane's 16
                // of the sample's 31 features, slot q = 8 ks + e <-> feature k = 16 ks + 8 h + e.  Feature order
                // [cano 3 | joints 24 | T-pose vertex 3 | distance 1] (legacy_mode: joints first, then cano).
                const float X = p[0], Y = p[1], Z = p[2];
                const int vi = a_nn[gi];
                const int64_t vrow = (int64_t)b * A.V + vi;
                float f[16];
                // every slot as a joint distance first: row k (+ 3 in legacy order) of the padded joint table, one broadcast
                // ds_read_b128 per slot; the non-joint slots are overwritten below
                const lds_ptr jb = lane_base(joint0 + opaque, (8 * h + (A.legacy_mode ? 3 : 0)) * 16);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 J = lds_ld<f32x4>(jb + ((q >> 3) * 16 + (q & 7)) * 16);
                    const float ax = X - J.x, ay = Y - J.y, az = Z - J.z;
                    // v_sqrt_f32 (1 ulp) and a reciprocal constant instead of the IEEE sqrt / divide sequences (~10 instructions
                    // each, 16 times per lane and step); the features differ from h3d_geo_features' in the last bit at most
                    f[q] = __builtin_amdgcn_sqrtf(ax * ax + ay * ay + az * az) * (1.f / 2.4f);
                }
                if (h == (A.legacy_mode ? 1 : 0)) {            // canonical coordinates: features 0..2 (legacy: 24..26)
                    const __attribute__((address_space(1))) f32x4* M = reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(a_vik + vrow * 16);
                    const f32x4 r0 = M[0], r1 = M[1], r2 = M[2];
                    const float cx = (r0.x * X + r0.y * Y + r0.z * Z + r0.w) * 0.5f;
                    const float cy = ((r1.x * X + r1.y * Y + r1.z * Z + r1.w) + 0.2f) * 0.5f;
                    const float cz = (r2.x * X + r2.y * Y + r2.z * Z + r2.w) * (1.f / 1.3f);
                    if (A.legacy_mode) { f[8] = cx; f[9] = cy; f[10] = cz; }
                    else { f[0] = cx; f[1] = cy; f[2] = cz; }
                }
                if (h == 1) {                                   // features 27..30 (+ the padding slot 31)
                    gcf tv = a_tpose + vrow * 3;
                    gcf vv = a_verts + vrow * 3;
                    f[11] = tv[0]; f[12] = tv[1]; f[13] = tv[2] * (1.f / 0.2f);
                    f[14] = __builtin_amdgcn_sqrtf(sqdist_exact(X, Y, Z, vv[0], vv[1], vv[2])) * (1.f / 1.3f);
                    f[15] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ok ? f[ks * 8 + e] : 0.f;
                    split8(v, kSIn, gh[ks], gl[ks]);
                }
            }
        }
        // ---- the layer chain.  Producers (FiLM epilogue of a layer's source accumulators) are created up front because the
        //      tile-0 chunks of producer l+1 run inside the last k-step of GEMM l.
        FilmProducer<NT, X2> p_coord{Y, xh, xl, b6, lane_base(tab + ST_COORD * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_geo{Y, xh, xl, b6, lane_base(tab + ST_GEO * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_f0{X, xh, xl, b6, lane_base(tab + ST_FILM0 * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_f1{Y, xh, xl, b6, lane_base(tab + ST_FILM1 * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_f2{X, xh, xl, b6, lane_base(tab + ST_FILM2 * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_f3{Y, xh, xl, b6, lane_base(tab + ST_FILM3 * 2 * HdP, 32 * h)};
        FilmProducer<NT, X2> p_col{X, xh, xl, b6, lane_base(tab + ST_COLOR * 2 * HdP, 32 * h)};
        // coordinate first layer -> Y ; FiLM 0, coordinate half: X = W0a * sin(30 * (Wc p + bc))
        {
            const half8 ih[2] = {ch, ch}, il[2] = {cl, cl};
            input_layer<NT, 1>(Y, ih, il, ring, p_coord);
        }
        pin_agpr<NT>(Y);
        layer<NT, KS, true, false, false, X2>(X, xh, xl, b6, ring, p_coord, none, hacc, head_lds, lane);
        pin_agpr<NT>(X);
        // geometry first layer -> Y ; FiLM 0, geometry half: X += W0b * sin(30 * (Wg g + bg))
        input_layer<NT, 2>(Y, gh, gl, ring, p_geo);
        pin_agpr<NT>(X); pin_agpr<NT>(Y);
        layer<NT, KS, false, false, false, X2>(X, xh, xl, b6, ring, p_geo, p_f0, hacc, head_lds, lane);
        pin_agpr<NT>(X);
        // FiLM 1..3
        layer<NT, KS, true, false, false, X2>(Y, xh, xl, b6, ring, p_f0, p_f1, hacc, head_lds, lane);
        pin_agpr<NT>(Y);
        layer<NT, KS, true, false, false, X2>(X, xh, xl, b6, ring, p_f1, p_f2, hacc, head_lds, lane);
        pin_agpr<NT>(X);
        layer<NT, KS, true, false, false, X2>(Y, xh, xl, b6, ring, p_f2, p_f3, hacc, head_lds, lane);
        pin_agpr<NT>(Y);
        // colour FiLM: X = Wc[:, 3:] * film3(Y) + Wc[:, :3] * dir  (+ density head on film3(Y))
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (h == 0) {
                if (a_dirs) {
                    if (ok) { v[0] = a_dirs[gi * 3]; v[1] = a_dirs[gi * 3 + 1]; v[2] = a_dirs[gi * 3 + 2]; }
                } else {
                    v[2] = -1.f;                      // lock_view_dependence: (0, 0, -1)
                }
            }
            split8(v, kSA, xh[KS], xl[KS]);
        }
        layer<NT, KS + 1, true, false, true, X2>(X, xh, xl, b6, ring, p_f3, p_col, hacc, head_lds, lane);
        pin_agpr<NT>(X);
        H3D_TRACE(7);
        // density of this lane's sample: head row 0 = accumulator register 0 of the lower lane half
        const float sigma = __shfl(hacc[0], m, 64) * hi0 + hb0;
        float w = 0.f, bg = 0.f;
        if (!FUSED) {
            if (ok && h == 0) a_out[gi * (F + 4) + F + 3] = sigma;
        } else {
            // compositing weights of these 32 samples (both lane halves compute identical values)
            const int s_idx = A.log2S < 0 ? si * 32 + m : (int)(n & (S - 1));
            float alpha = 0.f, f = 1.f, z = 0.f;
            float last_abs = -1.f;                          // |density| of a ray's last sample held by this lane (ref_mode 1), else -1
            if (ok) {
                z = a_z[gi];
                const float delta = (s_idx == S - 1) ? 1e9f : a_z[gi + 1] - z;
                const float sgn = sigma + (a_noise ? a_noise[gi] : 0.f);
                alpha = 1.f - expf(-delta * density(sgn, A.clamp_mode));
                f = (1.f - alpha) + 1e-12f;
                if (ref_mode == 1) {
                    smax = fmaxf(smax, fabsf(sgn));
                    if (s_idx == S - 1) last_abs = fabsf(sgn);
                }
            }
            const int sl = m & (seglen - 1);
            float incl = f;
            for (int off = 1; off < seglen; off <<= 1) {
                const float u = __shfl_up(incl, off, 32);
                if (sl >= off) incl *= u;
            }
            float excl = __shfl_up(incl, 1, 32);
            if (sl == 0) excl = 1.f;
            w = alpha * (carryT * excl);
            float wsum = w, dsum = w * z;
            for (int off = seglen >> 1; off > 0; off >>= 1) {
                wsum += __shfl_xor(wsum, off, 32);
                dsum += __shfl_xor(dsum, off, 32);
            }
            const float z_last = __shfl(z, m | (seglen - 1), 32);
            carryT *= __shfl(incl, 31, 32);
            carryW += wsum;
            carryD += dsum;
            if (last_step) {
                bg = 1.f - carryW;
                if (ok && s_idx == S - 1) {
                    if (h == 0) a_depth[ray_of(n)] = carryD + bg * z_last;
                    if (A.last_back) w += bg;
                }
            }
            if (ok && h == 0) a_weights[gi] = w;
            if (h == 0) { wl_lds[m] = w; wl_lds[32 + m] = bg; }
            if (ref_mode == 1 && last_step) {
                // the largest |density| of each ray of this unit (all steps), floored by ref_scale; a unit with a last sample
                // inside ref_eps of that scale is listed for the refinement launch (one list entry per unit, written by lane 0)
                float rmax = smax;
                for (int off = 1; off < seglen; off <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, off, 32));
                const bool ill = last_abs >= 0.f && last_abs <= ref_eps * fmaxf(rmax, ref_scale);
                if (__any(ill)) {
                    if (lane == 0) {
                        // (an address-space-1 pointer: a GLOBAL atomic -- through a generic pointer it would be a FLAT access, see the
                        // note on the rgb stores in synthesis_x3.hip; the wait behind it sits on this rare path only)
                        const int slot = __hip_atomic_fetch_add(ref_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (slot < ref_cap) ref_list[slot] = (int)uidx;
                    }
                }
            }
        }

        // ---- feature head, sample-major accumulator: Y = film_color(X)^T * Wf^T  (+ colour heads on film_color(X))
        layer<NT, KS, true, true, true, X2>(Y, xh, xl, b6, ring, p_col, none, hacc, head_lds, lane);
        pin_agpr<NT>(Y);
        float rgb[3];
        {
            const float c0 = __shfl(hacc[1], m, 64), c1 = __shfl(hacc[2], m, 64), c2 = __shfl(hacc[3], m, 64);
            rgb[0] = 1.f / (1.f + expf(-(c0 * hi1 + hb1)));
            rgb[1] = 1.f / (1.f + expf(-(c1 * hi2 + hb2)));
            rgb[2] = 1.f / (1.f + expf(-(c2 * hi3 + hb3)));
        }
        f32x16 (&acc)[NT] = Y;
        H3D_TRACE(5);
        if (!FUSED) {
            if (ok && h == 0) {
                a_out[gi * (F + 4) + 0] = rgb[0];
                a_out[gi * (F + 4) + 1] = rgb[1];
                a_out[gi * (F + 4) + 2] = rgb[2];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nn = nt * 32 + m;
                if (nn >= F) continue;
                const float bias = tfeat[nn];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r >> 2) * 8 + 4 * h + (r & 3);
                    const int64_t pn = n0 + row;
                    if (pn < N) a_out[((int64_t)b * N + pn) * (F + 4) + 3 + nn] = fmaf(acc[nt][r], inv_f, bias);
                }
            }
        } else {
            const int C = F + 3;
            // colour: weighted sum over the lanes of each ray
            {
                float v[3] = {w * rgb[0], w * rgb[1], w * rgb[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    for (int off = seglen >> 1; off > 0; off >>= 1) v[c] += __shfl_xor(v[c], off, 32);
                // lanes 0..2 of each segment (h == 0) own one colour channel
                const int sl = m & (seglen - 1);
                float mine = sl == 0 ? v[0] : sl == 1 ? v[1] : v[2];
                if (S > 32) {                      // one ray per wave: accumulate over its steps
                    rgbacc += mine;
                    mine = rgbacc;
                }
                if (last_step && h == 0 && sl < 3 && ok) {
                    const int64_t ray = ray_of(n);
                    a_feats[ray * C + sl] = mine + (A.white_back ? bg : 0.f);
                }
            }
            // features: sum over the rows (samples) of each ray held in this lane's accumulator registers
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // sum_r w_r * (acc_r * inv + bias) = inv * sum_r w_r acc_r + bias * sum_r w_r
            float wr[16], wsum[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 w4 = ld4(wl_lds + rg * 8 + 4 * h);
                wr[rg * 4 + 0] = w4.x; wr[rg * 4 + 1] = w4.y; wr[rg * 4 + 2] = w4.z; wr[rg * 4 + 3] = w4.w;
                wsum[rg] = (w4.x + w4.y) + (w4.z + w4.w);
            }
            const float wtot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            const float back = A.white_back ? wl_lds[32] : 0.f;
            // per-lane output addresses and background terms, once per step: the tile loop below then needs no scalar pointer,
            // ray index or kernel argument (they were spilled SGPRs, reloaded with v_readlane in every tile iteration)
            gf fout = a_feats + ray_of(n0) * C + 3 + m;                       // S >= 32: one ray per wave step
            gf frg[4];
            float brg[4];
            bool okrg[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {                                      // S < 32: a ray per 8 / 16 rows
                const int64_t n_first = n0 + rg * 8;
                okrg[rg] = h == 0 && n_first < N;
                frg[rg] = a_feats + ray_of(okrg[rg] ? n_first : n0) * C + 3 + m;
                brg[rg] = A.white_back ? wl_lds[32 + rg * 8] : 0.f;
            }
            const bool store_ray = last_step && h == 0 && n0 < N;
            const int g8 = S >> 3;                 // 8-row groups per ray when S < 32: 1 or 2
            const bool long_rays = S >= 32;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nn = nt * 32 + m;
                const bool okn = nn < F;
                const float bias = tfeat[okn ? nn : 0];
                pin1(acc[nt]);
                float s4[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float s = acc[nt][rg * 4 + 0] * wr[rg * 4 + 0];
                    s = fmaf(acc[nt][rg * 4 + 1], wr[rg * 4 + 1], s);
                    s = fmaf(acc[nt][rg * 4 + 2], wr[rg * 4 + 2], s);
                    s4[rg] = fmaf(acc[nt][rg * 4 + 3], wr[rg * 4 + 3], s);
                }
                if (long_rays) {
                    float tot = fmaf((s4[0] + s4[1]) + (s4[2] + s4[3]), inv_f, bias * wtot);
                    tot += __shfl_xor(tot, 32, 64);
                    rayacc[nt] += tot;
                    if (store_ray && okn) fout[nt * 32] = rayacc[nt] + back;
                } else {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        if (rg % g8 != 0) continue;
                        float sv = fmaf(s4[rg], inv_f, bias * wsum[rg]);
                        if (g8 == 2) sv += fmaf(s4[rg + 1 < 4 ? rg + 1 : 3], inv_f, bias * wsum[rg + 1 < 4 ? rg + 1 : 3]);
                        sv += __shfl_xor(sv, 32, 64);
                        if (okn && okrg[rg]) frg[rg][nt * 32] = sv + brg[rg];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    }   // unit groups
    ring.drain();
    H3D_TRACE(9);
    H3D_TRACE_DUMP(A.out);
}

size_t lds_bytes(const LayoutX3& L, bool geoin = false) {
    return sizeof(float) * ((size_t)ST_COUNT * 2 * L.HdP + L.HdP + 4 * 64 + (geoin ? kJointRows * 4 : 0)) +
           (size_t)4 * (L.head_planes * L.KS * 32 + kHeadPad) + (L.head_planes == 3 ? kRingX2 : H3D_RING_DEPTH) * (size_t)L.NT * 2048;
}

template <int NT, bool FUSED, bool X2, bool GEOIN = false>
int launch_one(Args A, int B, int64_t groups, hipStream_t st) {
    H3D_ALLOW_MAX_LDS((field_x3_kernel<NT, FUSED, X2, GEOIN>));
    A.n_groups = (int)groups;
    // about four persistent workgroups per CU in total (one resident per CU: registers): tables once per many unit groups, short tail
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
    }
    static const int per_cu = getenv("H3D_FIELD_WG_PER_CU") ? atoi(getenv("H3D_FIELD_WG_PER_CU")) : 4;      // 0: one unit group per workgroup
    int64_t per_sample = per_cu <= 0 ? groups : std::max<int64_t>(1, std::min<int64_t>(groups, ((int64_t)per_cu * cus + B - 1) / B));
    // refinement launch: a handful of listed units per batch item (workgroups beyond the list leave at once)
    if (A.ref_mode == 2) per_sample = std::max<int64_t>(1, std::min<int64_t>((A.ref_cap + 3) / 4, 8));
    h3d::pre_launch();
    hipLaunchKernelGGL((field_x3_kernel<NT, FUSED, X2, GEOIN>), dim3((unsigned)per_sample, (unsigned)B), dim3(256), lds_bytes(A.L, GEOIN), st, A);
    return h3d::launch_status(FUSED ? (X2 ? "h3d_render_fused_x2" : "h3d_render_fused_x3") : (X2 ? "h3d_neural_field_x2" : "h3d_neural_field_x3"));
}

template <bool FUSED>
int launch(const Args& A, int B, int64_t groups, hipStream_t st) {
    if constexpr (FUSED) {
        if (A.nn_index) {            // A4 inside the kernel
            const bool x2 = A.L.head_planes == 3;
            if (A.L.NT == 4) return x2 ? launch_one<4, true, true, true>(A, B, groups, st) : launch_one<4, true, false, true>(A, B, groups, st);
            if (A.L.NT == 8) return x2 ? launch_one<8, true, true, true>(A, B, groups, st) : launch_one<8, true, false, true>(A, B, groups, st);
        }
    }
    if (A.L.head_planes == 3) {
        switch (A.L.NT) {
            case 4: return launch_one<4, FUSED, true>(A, B, groups, st);
            case 8: return launch_one<8, FUSED, true>(A, B, groups, st);
            default: break;
        }
    }
    switch (A.L.NT) {
        case 4: return launch_one<4, FUSED, false>(A, B, groups, st);
        case 8: return launch_one<8, FUSED, false>(A, B, groups, st);
        default:
            h3d::set_error("x3 field kernel: width %d exceeds the 256 this engine keeps in registers "
                           "(use the fp32 engine, h3d_neural_field / h3d_render_fused)", A.L.HdP);
            return H3D_EUNSUPPORTED;
    }
}

// ---------------------------------------------------------------- host-side packing

__host__ __device__ inline uint16_t f32_to_f16_rn(float f) {            // round-to-nearest-even, handles subnormals; inputs are finite
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7bffu);          // clamp to max finite (never hit: scaled)
    if (x < 0x38800000u) {                                             // subnormal / zero in f16
        if (x < 0x33000000u) return (uint16_t)sign;
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - (int)(x >> 23);                       // 14..24
        uint32_t r = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}

__host__ __device__ inline float f16_to_f32(uint16_t v) {
    const uint32_t sign = (uint32_t)(v & 0x8000u) << 16;
    uint32_t e = (v >> 10) & 0x1f, m = v & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; ++s; }
            x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13);
        }
    } else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// floor(log2(q)) of a positive finite q, from the exponent field (exact; log2f may round up to the next integer just below a
// power of two, and host and device must agree bit for bit: h3d_field_pack_*_device)
__host__ __device__ inline int floor_log2(float q) {
    uint32_t x;
    memcpy(&x, &q, 4);
    const int e = (int)((x >> 23) & 0xffu);
    if (e) return e - 127;
    int s = 0;                                   // subnormal
    for (uint32_t m = x & 0x7fffffu; m && !(m & 0x400000u); m <<= 1) ++s;
    return -127 - s;
}
// the largest power of two 2^e with mx * 2^e <= target (1 for an all-zero matrix)
__host__ __device__ inline float pow2_scale_of(float mx, float target) { return mx > 0.f ? ldexpf(1.f, floor_log2(target / mx)) : 1.f; }

float pow2_scale(const float* w, int64_t n, float target) {
    float mx = 0.f;
    for (int64_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    return pow2_scale_of(mx, target);
}

// input feature of k-slot (half hh, element e) of k-step ks when the consumer's B fragments are accumulator registers
// (see FilmProducer): tile ks/2, accumulator register r = 8*(ks & 1) + e  ->  row (r & 3) + 8*(r >> 2) + 4*hh
__host__ __device__ inline int acc_k(int ks, int hh, int e) { return 32 * (ks / 2) + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hh; }

// W [n_out, ld] row-major; K range [in_begin, in_begin+in_count) -> [KSm][NT][2][64][8] f16, scaled by `scale`.
// acc_order: K runs in accumulator-register order (the input comes from a previous layer's accumulators) instead of
// the natural order (inputs assembled from memory: coordinates, geometry features, view direction).
__host__ __device__ inline void pack_x3_unit(const float* w, int ld, int in_begin, int in_count, int n_out, int NT, float scale,
                                             uint16_t* dst, bool acc_order, int ks, int nt, int lane) {
    for (int e = 0; e < 8; ++e) {
        const int k = acc_order ? acc_k(ks, lane >> 5, e) : 16 * ks + 8 * (lane >> 5) + e, nn = 32 * nt + (lane & 31);
        float v = 0.f;
        if (k < in_count && nn < n_out) v = w[(int64_t)nn * ld + in_begin + k] * scale;
        const uint16_t hi = f32_to_f16_rn(v);
        const uint16_t lo = f32_to_f16_rn(v - f16_to_f32(hi));
        const int64_t base = (((int64_t)ks * NT + nt) * 2) * 64 * 8;
        dst[base + lane * 8 + e] = hi;
        dst[base + 64 * 8 + lane * 8 + e] = lo;
    }
}
void pack_x3(const float* w, int ld, int in_begin, int in_count, int n_out, int KSm, int NT, float scale, uint16_t* dst,
             bool acc_order) {
    for (int ks = 0; ks < KSm; ++ks)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) pack_x3_unit(w, ld, in_begin, in_count, n_out, NT, scale, dst, acc_order, ks, nt, lane);
}

// ---- x2 packing: fp6 (e2m3) codes and block scales
__host__ __device__ inline float e2m3_value(unsigned c) {
    const int e = (c >> 3) & 3, m = c & 7;
    const float r = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return (c & 32) ? -r : r;
}
__host__ __device__ inline unsigned e2m3_code(float v) {            // round-to-nearest-even on the code grid, saturating at 7.5
    const unsigned sign = v < 0.f ? 32u : 0u;
    const float a = fminf(fabsf(v), 7.5f);
    const float step = a < 2.f ? 0.125f : a < 4.f ? 0.25f : 0.5f;
    const float q = nearbyintf(a / step) * step;         // default rounding mode: ties to even; spacing doubles exactly at 2 and 4
    unsigned c;
    if (q < 2.f) c = (unsigned)(q * 8.f);                // 0 .. 15: subnormals 0..7 and [1, 2)
    else if (q < 4.f) c = 16u + (unsigned)((q - 2.f) * 4.f);
    else c = 24u + (unsigned)((q - 4.f) * 2.f);
    return sign | c;
}

// W [n_out, ld] row-major; K range [in_begin, in_begin + in_count) in accumulator order over KSm (even) k-steps ->
// stages [KSm][NT][hi fragment 1 KiB | fp6 half-record 1 KiB], weights scaled by `scale` (the f16 scale of the matrix);
// the destination must be zero-initialised (the odd stages' half-records end in 256 B of padding)
__host__ __device__ inline void pack_x2_unit(const float* w, int ld, int in_begin, int in_count, int n_out, int NT, float scale,
                                             unsigned char* dst, int T, int nt, int lane) {
    const int nn = 32 * nt + (lane & 31), hh = lane >> 5;
    float hi[16], lo[16], mx = 0.f;
    for (int j = 0; j < 2; ++j)
        for (int e = 0; e < 8; ++e) {
            const int k = acc_k(2 * T + j, hh, e);
            float v = 0.f;
            if (k < in_count && nn < n_out) v = w[(int64_t)nn * ld + in_begin + k] * scale;
            const uint16_t h16 = f32_to_f16_rn(v);
            hi[8 * j + e] = f16_to_f32(h16);
            lo[8 * j + e] = v - hi[8 * j + e];
            mx = fmaxf(mx, fabsf(hi[8 * j + e]));
            uint16_t* hd = reinterpret_cast<uint16_t*>(dst + (((int64_t)(2 * T + j) * NT + nt) * 2) * 1024);
            hd[lane * 8 + e] = h16;
        }
    // block scale alpha = 2^ea: the largest with |hi| * alpha <= 7.5 unless a lo code would saturate (then half of
    // it); the instruction multiplies the codes by 2^(byte - 127) = 1 / alpha
    int ea = mx > 0.f ? floor_log2(7.5f / mx) : 0;
    if (ea > 100) ea = 100;
    if (ea < -100) ea = -100;
    // (f16-subnormal hi values leave lo up to 2^-1 of hi instead of 2^-11: several steps then)
    for (bool sat = true; sat && ea > -100;) {
        sat = false;
        for (int i = 0; i < 16; ++i) sat = sat || fabsf(lo[i]) * kX2Rho * ldexpf(1.f, ea) > 7.5f;
        if (sat) --ea;
    }
    const float alpha = ldexpf(1.f, ea);
    unsigned rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sl = 0; sl < 32; ++sl) {
        const float v = sl < 16 ? hi[sl] * alpha : lo[sl - 16] * alpha * kX2Rho;
        const uint64_t code = e2m3_code(v);
        const int bit = 6 * sl;
        rec[bit / 32] |= (unsigned)(code << (bit & 31));
        if ((bit & 31) > 26) rec[bit / 32 + 1] |= (unsigned)(code >> (32 - (bit & 31)));
    }
    rec[6] = (unsigned)(127 - ea) * 0x01010101u;
    // even stage: code dwords 0-3, 16 B per lane; odd stage: code dwords 4-5 as [64 lanes][8 B], then the scale dwords as
    // [64 lanes][4 B], then 256 B of zeros -- dense, so that gemm_x2_roll's 64- and 32-bit reads are bank-conflict-free
    unsigned* ev = reinterpret_cast<unsigned*>(dst + (((int64_t)(2 * T) * NT + nt) * 2 + 1) * 1024);
    unsigned* od = reinterpret_cast<unsigned*>(dst + (((int64_t)(2 * T + 1) * NT + nt) * 2 + 1) * 1024);
    for (int d = 0; d < 4; ++d) ev[lane * 4 + d] = rec[d];
    od[lane * 2 + 0] = rec[4];
    od[lane * 2 + 1] = rec[5];
    od[128 + lane] = rec[6];
}
void pack_x2(const float* w, int ld, int in_begin, int in_count, int n_out, int KSm, int NT, float scale, unsigned char* dst) {
    for (int T = 0; T < KSm / 2; ++T)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) pack_x2_unit(w, ld, in_begin, in_count, n_out, NT, scale, dst, T, nt, lane);
}

}  // namespace

extern "C" int h3d_field_x3_layout(int Hd, int F, int64_t* out, int n_out) {
    H3D_REQUIRE(out && n_out >= 20, "h3d_field_x3_layout: need room for 20 values");
    H3D_REQUIRE(Hd >= 1 && F >= 1 && Hd <= 256 && F <= 256, "h3d_field_x3_layout: widths up to 256 (got %d, %d)", Hd, F);
    const LayoutX3 L = make_layout(Hd, F);
    int i = 0;
    out[i++] = L.NT; out[i++] = L.KS; out[i++] = L.HdP; out[i++] = L.stages;
    for (int w = 0; w < W_COUNT; ++w) out[i++] = L.w[w];
    out[i++] = L.inv_scale; out[i++] = L.bias; out[i++] = L.b_feat; out[i++] = L.head_w; out[i++] = L.head_inv;
    out[i++] = L.head_b; out[i++] = L.total;
    return H3D_OK;
}

extern "C" int64_t h3d_field_pack_x3_size(int Hd, int F) {
    if (Hd < 1 || F < 1 || Hd > 256 || F > 256) return -1;
    return make_layout(Hd, F).total;
}

static int field_pack(const h3d_field_params* p, int Hd, int F, void* blob_, bool x2) {
    H3D_REQUIRE(p && blob_, "h3d_field_pack_x3 / _x2: null pointer");
    H3D_REQUIRE(Hd >= 1 && F >= 1 && Hd <= 256 && F <= 256, "h3d_field_pack_x3 / _x2: widths up to 256 (got %d, %d)", Hd, F);
    const LayoutX3 L = make_layout(Hd, F, x2);
    unsigned char* blob = static_cast<unsigned char*>(blob_);
    memset(blob, 0, L.total);
    float* invs = reinterpret_cast<float*>(blob + L.inv_scale);
    const float target = 8192.f;
    auto mat = [&](int wi, const float* w, int ld, int in_begin, int in_count, int n_out, int KSm, float in_scale, bool acc_order) {
        // scale taken over the slice actually used
        float mx = 0.f;
        for (int nn = 0; nn < n_out; ++nn)
            for (int k = 0; k < in_count; ++k) mx = fmaxf(mx, fabsf(w[(int64_t)nn * ld + in_begin + k]));
        const float sc = pow2_scale_of(mx, target);
        if (x2 && acc_order) pack_x2(w, ld, in_begin, in_count, n_out, KSm, L.NT, sc, blob + L.w[wi]);
        else pack_x3(w, ld, in_begin, in_count, n_out, KSm, L.NT, sc, reinterpret_cast<uint16_t*>(blob + L.w[wi]), acc_order);
        invs[wi] = 1.f / (sc * in_scale);
        return sc;
    };
    mat(W_COORD, p->w_coord, 3, 0, 3, Hd, 1, kSIn, false);
    mat(W_GEO, p->w_geo, 31, 0, 31, Hd, 2, kSIn, false);
    // FiLM 0: both halves must share one scale because they accumulate into the same registers
    {
        const float sc = pow2_scale(p->w_film[0], (int64_t)Hd * 2 * Hd, target);
        if (x2) {
            pack_x2(p->w_film[0], 2 * Hd, 0, Hd, Hd, L.KS, L.NT, sc, blob + L.w[W_F0A]);
            pack_x2(p->w_film[0], 2 * Hd, Hd, Hd, Hd, L.KS, L.NT, sc, blob + L.w[W_F0B]);
        } else {
            pack_x3(p->w_film[0], 2 * Hd, 0, Hd, Hd, L.KS, L.NT, sc, reinterpret_cast<uint16_t*>(blob + L.w[W_F0A]), true);
            pack_x3(p->w_film[0], 2 * Hd, Hd, Hd, Hd, L.KS, L.NT, sc, reinterpret_cast<uint16_t*>(blob + L.w[W_F0B]), true);
        }
        invs[W_F0A] = invs[W_F0B] = 1.f / (sc * kSA);
    }
    for (int l = 1; l < 4; ++l) mat(W_F1 + l - 1, p->w_film[l], Hd, 0, Hd, Hd, L.KS, kSA, true);
    // colour layer: KS k-steps over the hidden features (columns 3..) + one k-step over the view direction (columns 0..2),
    // one scale for the whole matrix (same accumulators)
    {
        const float sc = pow2_scale(p->w_color, (int64_t)Hd * (Hd + 3), target);
        uint16_t* dst = reinterpret_cast<uint16_t*>(blob + L.w[W_COLOR]);
        if (x2) pack_x2(p->w_color, Hd + 3, 3, Hd, Hd, L.KS, L.NT, sc, blob + L.w[W_COLOR]);
        else pack_x3(p->w_color, Hd + 3, 3, Hd, Hd, L.KS, L.NT, sc, dst, true);
        pack_x3(p->w_color, Hd + 3, 0, 3, Hd, 1, L.NT, sc, dst + (int64_t)L.KS * L.NT * 2 * 64 * 8, false);
        invs[W_COLOR] = 1.f / (sc * kSA);
    }
    mat(W_FEAT, p->w_feat, Hd, 0, Hd, F, L.KS, kSA, true);
    float* bias = reinterpret_cast<float*>(blob + L.bias);
    for (int nn = 0; nn < Hd; ++nn) {
        bias[ST_GEO * L.HdP + nn] = p->b_geo[nn];
        bias[ST_COORD * L.HdP + nn] = p->b_coord[nn];
        for (int l = 0; l < 4; ++l) bias[(ST_FILM0 + l) * L.HdP + nn] = p->b_film[l][nn];
        bias[ST_COLOR * L.HdP + nn] = p->b_color[nn];
    }
    float* bf = reinterpret_cast<float*>(blob + L.b_feat);
    for (int nn = 0; nn < F; ++nn) bf[nn] = p->b_feat[nn];
    // heads: [head][hi / lo (/ hi * 2^-12)][ks][half][8]
    const int PL = L.head_planes;
    uint16_t* hw = reinterpret_cast<uint16_t*>(blob + L.head_w);
    float* hinv = reinterpret_cast<float*>(blob + L.head_inv);
    float* hb = reinterpret_cast<float*>(blob + L.head_b);
    for (int hd = 0; hd < 4; ++hd) {
        const float* w = hd == 0 ? p->w_sigma : p->w_rgb + (int64_t)(hd - 1) * Hd;
        const float sc = pow2_scale(w, Hd, target);
        for (int ks = 0; ks < L.KS; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                for (int e = 0; e < 8; ++e) {
                    const int k = acc_k(ks, hh, e);
                    const float v = k < Hd ? w[k] * sc : 0.f;
                    const uint16_t hi = f32_to_f16_rn(v), lo = f32_to_f16_rn(v - f16_to_f32(hi));
                    hw[((((int64_t)hd * PL + 0) * L.KS + ks) * 2 + hh) * 8 + e] = hi;
                    hw[((((int64_t)hd * PL + 1) * L.KS + ks) * 2 + hh) * 8 + e] = lo;
                    if (x2) hw[((((int64_t)hd * PL + 2) * L.KS + ks) * 2 + hh) * 8 + e] = f32_to_f16_rn(f16_to_f32(hi) / kX2Rho);
                }
        hinv[hd] = 1.f / (sc * kSA);
        hb[hd] = hd == 0 ? p->b_sigma[0] : p->b_rgb[hd - 1];
    }
    return H3D_OK;
}


// ---------------------------------------------------------------- device-side packing (round 6)
// The same blob from parameters that live on the DEVICE, in one launch: weights that change every optimiser step (the D step's
// no-grad generator forward, /root/reference/lib/trainers/phase_trainer.py:355-362) reach the fused render without the D2H copy,
// host pack and H2D copy of field_pack (~10 ms and a stream synchronisation per weight version).  The arithmetic is the host
// packer's own (pack_x3_unit / pack_x2_unit / pow2_scale_of are __host__ __device__; every operation in them is exact or an
// IEEE-rounded division), so the blobs are bit-identical (tests/test_gpu_field_pack_device.py).
namespace {

struct PackJob {
    const float* w;                       // matrix to pack (rows of `ld` floats)
    int ld, in_begin, in_count, n_out, KSm;
    int kind;                             // 0: x3, natural K order; 1: x3, accumulator K order; 2: x2 (accumulator order)
    int64_t dst;                          // byte offset of the matrix in the blob
    int s_rows, s_ld, s_begin, s_count;   // the slice of `w` the scale is taken over (rows x columns)
    int inv_slot;                         // index into inv_scale[] (or -1)
    float in_scale;
};
constexpr int kPackJobs = 10, kPackSplit = 8;
struct PackArgs {
    PackJob job[kPackJobs];
    h3d_field_params p;
    LayoutX3 L;
    unsigned char* blob;
    int Hd, F, x2;
};

__device__ inline float block_max(float v, float* red) {       // 256 threads; every thread returns the maximum
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);
        __syncthreads();
    }
    const float m = red[0];
    __syncthreads();
    return m;
}

__global__ __launch_bounds__(256) void field_pack_kernel(PackArgs A) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    const float target = 8192.f;
    const LayoutX3& L = A.L;
    unsigned char* blob = A.blob;
    if ((int)blockIdx.x == kPackJobs) {
        // biases and the four heads (one workgroup)
        if (blockIdx.y) return;
        float* bias = reinterpret_cast<float*>(blob + L.bias);
        for (int nn = t; nn < A.Hd; nn += 256) {
            bias[ST_GEO * L.HdP + nn] = A.p.b_geo[nn];
            bias[ST_COORD * L.HdP + nn] = A.p.b_coord[nn];
            for (int l = 0; l < 4; ++l) bias[(ST_FILM0 + l) * L.HdP + nn] = A.p.b_film[l][nn];
            bias[ST_COLOR * L.HdP + nn] = A.p.b_color[nn];
        }
        float* bf = reinterpret_cast<float*>(blob + L.b_feat);
        for (int nn = t; nn < A.F; nn += 256) bf[nn] = A.p.b_feat[nn];
        const int PL = L.head_planes;
        uint16_t* hw = reinterpret_cast<uint16_t*>(blob + L.head_w);
        float* hinv = reinterpret_cast<float*>(blob + L.head_inv);
        float* hb = reinterpret_cast<float*>(blob + L.head_b);
        for (int hd = 0; hd < 4; ++hd) {
            const float* w = hd == 0 ? A.p.w_sigma : A.p.w_rgb + (int64_t)(hd - 1) * A.Hd;
            float mx = 0.f;
            for (int k = t; k < A.Hd; k += 256) mx = fmaxf(mx, fabsf(w[k]));
            const float sc = pow2_scale_of(block_max(mx, red), target);
            for (int u = t; u < L.KS * 16; u += 256) {
                const int ks = u >> 4, hh = (u >> 3) & 1, e = u & 7;
                const int k = acc_k(ks, hh, e);
                const float v = k < A.Hd ? w[k] * sc : 0.f;
                const uint16_t hi = f32_to_f16_rn(v), lo = f32_to_f16_rn(v - f16_to_f32(hi));
                hw[((((int64_t)hd * PL + 0) * L.KS + ks) * 2 + hh) * 8 + e] = hi;
                hw[((((int64_t)hd * PL + 1) * L.KS + ks) * 2 + hh) * 8 + e] = lo;
                if (A.x2) hw[((((int64_t)hd * PL + 2) * L.KS + ks) * 2 + hh) * 8 + e] = f32_to_f16_rn(f16_to_f32(hi) / kX2Rho);
            }
            if (t == 0) {
                hinv[hd] = 1.f / (sc * kSA);
                hb[hd] = hd == 0 ? A.p.b_sigma[0] : A.p.b_rgb[hd - 1];
            }
        }
        return;
    }
    const PackJob& J = A.job[blockIdx.x];
    // every workgroup of a matrix takes the scale itself (<= 66 k values from the L2) and packs its share of the units
    float mx = 0.f;
    for (int i = t; i < J.s_rows * J.s_count; i += 256) {
        const int r = i / J.s_count, c = i - r * J.s_count;
        mx = fmaxf(mx, fabsf(J.w[(int64_t)r * J.s_ld + J.s_begin + c]));
    }
    const float sc = pow2_scale_of(block_max(mx, red), target);
    if (t == 0 && blockIdx.y == 0 && J.inv_slot >= 0) reinterpret_cast<float*>(blob + L.inv_scale)[J.inv_slot] = 1.f / (sc * J.in_scale);
    const int rows = J.kind == 2 ? J.KSm / 2 : J.KSm;               // k-steps (x3) or k-step pairs (x2)
    const int units = rows * L.NT * 64;
    for (int u = (int)blockIdx.y * 256 + t; u < units; u += kPackSplit * 256) {
        const int lane = u & 63, nt = (u >> 6) % L.NT, r = (u >> 6) / L.NT;
        if (J.kind == 2) pack_x2_unit(J.w, J.ld, J.in_begin, J.in_count, J.n_out, L.NT, sc, blob + J.dst, r, nt, lane);
        else pack_x3_unit(J.w, J.ld, J.in_begin, J.in_count, J.n_out, L.NT, sc, reinterpret_cast<uint16_t*>(blob + J.dst), J.kind == 1, r, nt, lane);
    }
}

int field_pack_device(const h3d_field_params* p, int Hd, int F, void* blob_, bool x2, h3d_stream_t stream) {
    H3D_REQUIRE(p && blob_, "h3d_field_pack_x3_device / _x2_device: null pointer");
    H3D_REQUIRE(Hd >= 1 && F >= 1 && Hd <= 256 && F <= 256, "h3d_field_pack_x3_device / _x2_device: widths up to 256 (got %d, %d)", Hd, F);
    H3D_REQUIRE(h3d::aligned16(blob_), "h3d_field_pack_x3_device / _x2_device: the blob must be 16-byte aligned");
    H3D_REQUIRE(p->w_coord && p->b_coord && p->w_geo && p->b_geo && p->w_sigma && p->b_sigma && p->w_color && p->b_color && p->w_rgb &&
                p->b_rgb && p->w_feat && p->b_feat, "h3d_field_pack_x3_device / _x2_device: null parameter pointer");
    for (int l = 0; l < 4; ++l) H3D_REQUIRE(p->w_film[l] && p->b_film[l], "h3d_field_pack_x3_device / _x2_device: null FiLM parameter pointer");
    PackArgs A{};
    A.L = make_layout(Hd, F, x2);
    A.p = *p; A.blob = static_cast<unsigned char*>(blob_); A.Hd = Hd; A.F = F; A.x2 = x2 ? 1 : 0;
    const LayoutX3& L = A.L;
    int n = 0;
    auto job = [&](int64_t dst, const float* w, int ld, int in_begin, int in_count, int n_out, int KSm, int kind, int s_rows, int s_begin,
                   int s_count, int inv_slot, float in_scale) {
        A.job[n++] = PackJob{w, ld, in_begin, in_count, n_out, KSm, kind, dst, s_rows, ld, s_begin, s_count, inv_slot, in_scale};
    };
    const int acc = x2 ? 2 : 1;
    job(L.w[W_COORD], p->w_coord, 3, 0, 3, Hd, 1, 0, Hd, 0, 3, W_COORD, kSIn);
    job(L.w[W_GEO], p->w_geo, 31, 0, 31, Hd, 2, 0, Hd, 0, 31, W_GEO, kSIn);
    job(L.w[W_F0A], p->w_film[0], 2 * Hd, 0, Hd, Hd, L.KS, acc, Hd, 0, 2 * Hd, W_F0A, kSA);      // both halves: one scale (same accumulators)
    job(L.w[W_F0B], p->w_film[0], 2 * Hd, Hd, Hd, Hd, L.KS, acc, Hd, 0, 2 * Hd, W_F0B, kSA);
    for (int l = 1; l < 4; ++l) job(L.w[W_F1 + l - 1], p->w_film[l], Hd, 0, Hd, Hd, L.KS, acc, Hd, 0, Hd, W_F1 + l - 1, kSA);
    job(L.w[W_COLOR], p->w_color, Hd + 3, 3, Hd, Hd, L.KS, acc, Hd, 0, Hd + 3, W_COLOR, kSA);
    job(L.w[W_COLOR] + (int64_t)L.KS * L.NT * 2 * 64 * 8 * 2, p->w_color, Hd + 3, 0, 3, Hd, 1, 0, Hd, 0, Hd + 3, -1, kSA);
    job(L.w[W_FEAT], p->w_feat, Hd, 0, Hd, F, L.KS, acc, F, 0, Hd, W_FEAT, kSA);
    if (n != kPackJobs) return H3D_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    h3d::pre_launch();
    if (hipMemsetAsync(A.blob, 0, (size_t)L.total, st) != hipSuccess) return h3d::launch_status("h3d_field_pack_device (memset)");
    hipLaunchKernelGGL(field_pack_kernel, dim3(kPackJobs + 1, kPackSplit), dim3(256), 0, st, A);
    return h3d::launch_status("h3d_field_pack_device");
}

}  // namespace

extern "C" int h3d_field_pack_x3_device(const h3d_field_params* p, int Hd, int F, void* blob, h3d_stream_t stream) {
    return field_pack_device(p, Hd, F, blob, false, stream);
}
extern "C" int h3d_field_pack_x2_device(const h3d_field_params* p, int Hd, int F, void* blob, h3d_stream_t stream) {
    return field_pack_device(p, Hd, F, blob, true, stream);
}

extern "C" int h3d_field_pack_x3(const h3d_field_params* p, int Hd, int F, void* blob) { return field_pack(p, Hd, F, blob, false); }
extern "C" int h3d_field_pack_x2(const h3d_field_params* p, int Hd, int F, void* blob) { return field_pack(p, Hd, F, blob, true); }
extern "C" int64_t h3d_field_pack_x2_size(int Hd, int F) {
    if (Hd < 1 || F < 1 || Hd > 256 || F > 256) return -1;
    return make_layout(Hd, F, true).total;
}
extern "C" int h3d_field_x2_layout(int Hd, int F, int64_t* out, int n_out) {
    H3D_REQUIRE(out && n_out >= 20, "h3d_field_x2_layout: need room for 20 values");
    H3D_REQUIRE(Hd >= 1 && F >= 1 && Hd <= 256 && F <= 256, "h3d_field_x2_layout: widths up to 256 (got %d, %d)", Hd, F);
    const LayoutX3 L = make_layout(Hd, F, true);
    int i = 0;
    out[i++] = L.NT; out[i++] = L.KS; out[i++] = L.HdP; out[i++] = L.stages;
    for (int w = 0; w < W_COUNT; ++w) out[i++] = L.w[w];
    out[i++] = L.inv_scale; out[i++] = L.bias; out[i++] = L.b_feat; out[i++] = L.head_w; out[i++] = L.head_inv;
    out[i++] = L.head_b; out[i++] = L.total;
    return H3D_OK;
}

static int check_x3(const void* packed, const float* points, const float* geo, const float* freq, const float* phase,
                    int B, int64_t N, int Hd, int F, int geo_stride) {
    H3D_REQUIRE(packed && points && geo && freq && phase, "x3 field: null pointer");
    H3D_REQUIRE(h3d::aligned16(packed), "x3 field: packed weights must be 16-byte aligned");
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "x3 field: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(Hd >= 1 && F >= 1, "x3 field: bad widths");
    H3D_REQUIRE(geo_stride >= 31, "x3 field: geo_stride=%d must be >= 31", geo_stride);
    if (Hd > 256 || F > 256) {
        h3d::set_error("x3 field kernel: widths up to 256 (got %d/%d); use the fp32 engine", Hd, F);
        return H3D_EUNSUPPORTED;
    }
    return H3D_OK;
}

static int neural_field_x(bool x2, const void* packed, const float* points, const float* geo, const float* dirs,
                          const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                          int geo_stride, float input_scaler, h3d_stream_t stream) {
    int rc = check_x3(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(out, "h3d_neural_field_x3: null output");
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const unsigned char*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase; A.out = out;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = 32; A.input_scaler = input_scaler;
    A.L = make_layout(Hd, F, x2);
    const int64_t groups = (N + 127) / 128;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_neural_field_x3: N too large");
    return launch<false>(A, B, groups, static_cast<hipStream_t>(stream));
}

extern "C" int h3d_neural_field_x3(const void* packed, const float* points, const float* geo, const float* dirs,
                                   const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                                   int geo_stride, float input_scaler, h3d_stream_t stream) {
    return neural_field_x(false, packed, points, geo, dirs, freq, phase, out, B, N, Hd, F, geo_stride, input_scaler, stream);
}
extern "C" int h3d_neural_field_x2(const void* packed, const float* points, const float* geo, const float* dirs,
                                   const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                                   int geo_stride, float input_scaler, h3d_stream_t stream) {
    return neural_field_x(true, packed, points, geo, dirs, freq, phase, out, B, N, Hd, F, geo_stride, input_scaler, stream);
}

struct GeoIn {            // A4 inside the fused kernel (h3d_render_fused_x2_geo / _x3_geo)
    const int* nn_index;
    const float *joints, *vertices, *tpose, *vertex_ik;
    int V, legacy_mode;
};
struct RefIn {            // refinement of ill-conditioned last samples (Args::ref_*): mode 1 = list (x2), 2 = redo the listed units (x3)
    int mode;
    float eps, scale;
    int* list;
    int* count;
    int cap;
};

static int render_fused_x(bool x2, const void* packed, const float* points, const float* geo, const float* dirs,
                          const float* freq, const float* phase, const float* z_vals, const float* noise,
                          float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                          int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                          h3d_stream_t stream, const GeoIn* gin = nullptr, const RefIn* ref = nullptr) {
    const int64_t N = (int64_t)R * S;
    if (ref) {
        H3D_REQUIRE(ref->list && ref->count && ref->cap >= 1 && ref->cap <= 65536, "h3d_render_fused_x*: refinement list / count / cap=%d", ref->cap);
        H3D_REQUIRE(ref->mode == 2 || (ref->eps >= 0.f && ref->scale >= 0.f), "h3d_render_fused_x2_*_ref: eps and scale must be >= 0");
    }
    if (gin) {
        H3D_REQUIRE(gin->nn_index && gin->joints && gin->vertices && gin->tpose && gin->vertex_ik, "h3d_render_fused_x*_geo: null pointer");
        H3D_REQUIRE(gin->V >= 1 && h3d::aligned16(gin->vertex_ik), "h3d_render_fused_x*_geo: V=%d, vertex_ik must be 16-byte aligned", gin->V);
        geo = points;                // check_x3 wants a non-null geometry pointer; the kernel never reads it
        geo_stride = 31;
    }
    int rc = check_x3(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(z_vals && feats && depth && weights, "h3d_render_fused_x3: null pointer");
    H3D_REQUIRE(clamp_mode == 0 || clamp_mode == 1, "h3d_render_fused_x3: clamp_mode must be 0 (relu) or 1 (softplus)");
    H3D_REQUIRE(R >= 0 && S >= 1, "h3d_render_fused_x3: bad R=%d S=%d", R, S);
    const bool ok_s = (S >= 8 && S <= 32 && (S & (S - 1)) == 0) || (S > 32 && S % 32 == 0);
    if (!ok_s) {
        h3d::set_error("h3d_render_fused_x3: S=%d unsupported (needs 8, 16, 32 or a multiple of 32)", S);
        return H3D_EUNSUPPORTED;
    }
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const unsigned char*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase;
    A.z_vals = z_vals; A.noise = noise; A.feats = feats; A.depth = depth; A.weights = weights;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = S; A.input_scaler = input_scaler;
    A.clamp_mode = clamp_mode; A.last_back = last_back; A.white_back = white_back;
    A.R = R;
    A.log2S = -1;
    if (S <= 32) { A.log2S = 0; while ((1 << A.log2S) < S) ++A.log2S; }
    A.L = make_layout(Hd, F, x2);
    if (gin) {
        A.nn_index = gin->nn_index; A.joints = gin->joints; A.vertices = gin->vertices; A.tpose = gin->tpose;
        A.vertex_ik = gin->vertex_ik; A.V = gin->V; A.legacy_mode = gin->legacy_mode;
    }
    const int unit = S > 32 ? S : 32;
    const int64_t units = (N + unit - 1) / unit;
    const int64_t groups = (units + 3) / 4;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_render_fused_x3: too many rays");
    if (ref) {
        A.ref_mode = ref->mode; A.ref_eps = ref->eps; A.ref_scale = ref->scale; A.ref_list = ref->list; A.ref_count = ref->count;
        A.ref_cap = ref->cap;
        if (ref->mode == 1 && hipMemsetAsync(ref->count, 0, sizeof(int) * (size_t)B, static_cast<hipStream_t>(stream)) != hipSuccess) {
            h3d::set_error("h3d_render_fused_x2_*_ref: hipMemsetAsync failed");
            return H3D_ELAUNCH;
        }
    }
#ifdef H3D_EXPERIMENT_TRACE
    {   // development build: dump the cycle trace of workgroup (1000, 3) to $H3D_TRACE_FILE after every launch
        static unsigned long long* tb = nullptr;
        if (!tb) (void)hipMalloc(&tb, 4096 * 8);
        (void)hipMemset(tb, 0, 4096 * 8);
        A.out = reinterpret_cast<float*>(tb);
        const int rc2 = launch<true>(A, B, groups, static_cast<hipStream_t>(stream));
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
    return launch<true>(A, B, groups, static_cast<hipStream_t>(stream));
}

extern "C" int h3d_render_fused_x3(const void* packed, const float* points, const float* geo, const float* dirs,
                                   const float* freq, const float* phase, const float* z_vals, const float* noise,
                                   float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                                   int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                                   h3d_stream_t stream) {
    return render_fused_x(false, packed, points, geo, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F,
                          geo_stride, input_scaler, clamp_mode, last_back, white_back, stream);
}
extern "C" int h3d_render_fused_x2(const void* packed, const float* points, const float* geo, const float* dirs,
                                   const float* freq, const float* phase, const float* z_vals, const float* noise,
                                   float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                                   int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                                   h3d_stream_t stream) {
    return render_fused_x(true, packed, points, geo, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F,
                          geo_stride, input_scaler, clamp_mode, last_back, white_back, stream);
}

/* The fused render with A4 inside (round 4; north_star: "the ray-sample / MLP / alpha-composite loop is a fused kernel"): instead
 * of a [B, N, 31] feature tensor the kernel takes every sample's nearest-vertex index (h3d_nearest_vertex) and the pose tables
 * and builds the sample's geometry features in the prologue of its 32-sample step -- canonical coordinates through the gathered
 * blended inverse transform, the 24 joint distances, the nearest T-pose vertex and the distance to the nearest vertex
 * (lib/components/smpl.py:210-249) -- as the B fragments of the K = 31 input GEMM.  Same results as h3d_geo_features followed by
 * h3d_render_fused_x2 / _x3 up to the rounding of the feature arithmetic. */
extern "C" int h3d_render_fused_x2_geo(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                       const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                       int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                       const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                       int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                       int white_back, h3d_stream_t stream) {
    const GeoIn g{nn_index, joints, vertices, tpose_vertices, vertex_ik, V, legacy_mode};
    return render_fused_x(true, packed, points, nullptr, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F, 31,
                          input_scaler, clamp_mode, last_back, white_back, stream, &g);
}
extern "C" int h3d_render_fused_x3_geo(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                       const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                       int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                       const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                       int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                       int white_back, h3d_stream_t stream) {
    const GeoIn g{nn_index, joints, vertices, tpose_vertices, vertex_ik, V, legacy_mode};
    return render_fused_x(false, packed, points, nullptr, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F, 31,
                          input_scaler, clamp_mode, last_back, white_back, stream, &g);
}

/* Refinement of ill-conditioned last samples (round 6).  The reference gives the last sample of a ray delta = 1e9
 * (lib/generators/volume_rendering.py:21): its alpha is 0 or 1 by the SIGN of its density, and with white_back the background
 * term flips by the whole remaining transmittance -- so a ray whose last density lies within the ARITHMETIC's error of zero
 * comes out wrong by O(1).  The x2 arithmetic's density error (~1e-4 of the densities' scale) makes that ~100 times likelier than
 * fp32-class arithmetic does.  h3d_render_fused_x2_geo_ref is h3d_render_fused_x2_geo that also lists, per batch item, the wave
 * units (a unit = one ray when S > 32, else the 32 / S rays of 32 consecutive samples) holding a ray with
 *     |sigma_last| <= ref_eps * max(max_s |sigma_s|, ref_scale)
 * in ref_list[b][0 .. ref_cap) (unit indices, any order) and counts them in ref_count[b] (zeroed by the call on `stream`; it
 * keeps counting beyond ref_cap).  h3d_render_fused_x3_geo_units is h3d_render_fused_x3_geo (`packed` in the x3 format) restricted
 * to exactly the listed units of every item: it overwrites their feats / depth / weights with the three-product engine's, so that
 * those rays take the sign of an fp32-class density.  Back to back on one stream: no host synchronisation. */
extern "C" int h3d_render_fused_x2_geo_ref(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                           const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                           int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                           const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                           int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                           int white_back, float ref_eps, float ref_scale, int32_t* ref_list, int32_t* ref_count,
                                           int ref_cap, h3d_stream_t stream) {
    const GeoIn g{nn_index, joints, vertices, tpose_vertices, vertex_ik, V, legacy_mode};
    const RefIn r{1, ref_eps, ref_scale, ref_list, ref_count, ref_cap};
    return render_fused_x(true, packed, points, nullptr, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F, 31,
                          input_scaler, clamp_mode, last_back, white_back, stream, &g, &r);
}
extern "C" int h3d_render_fused_x3_geo_units(const void* packed, const float* points, const int32_t* nn_index, const float* joints,
                                             const float* vertices, const float* tpose_vertices, const float* vertex_ik, int V,
                                             int legacy_mode, const float* dirs, const float* freq, const float* phase,
                                             const float* z_vals, const float* noise, float* feats, float* depth, float* weights,
                                             int B, int R, int S, int Hd, int F, float input_scaler, int clamp_mode, int last_back,
                                             int white_back, const int32_t* ref_list, const int32_t* ref_count, int ref_cap,
                                             h3d_stream_t stream) {
    const GeoIn g{nn_index, joints, vertices, tpose_vertices, vertex_ik, V, legacy_mode};
    const RefIn r{2, 0.f, 0.f, const_cast<int32_t*>(ref_list), const_cast<int32_t*>(ref_count), ref_cap};
    return render_fused_x(false, packed, points, nullptr, dirs, freq, phase, z_vals, noise, feats, depth, weights, B, R, S, Hd, F, 31,
                          input_scaler, clamp_mode, last_back, white_back, stream, &g, &r);
}
