// A7 + A8 + A9 on the bf16 matrix cores with split ("x3") operands, for gfx950.
//
// Same network, same exact host-side folding and same descriptor as synthesis.hip (see there for the reference
// citations and the algebra); different engine, shared with field_x3.hip (x3_common.hpp):
//   * a wavefront owns 32 pixels and all C channels; activations stay in registers from the coordinate input to
//     the last ToRGB (accumulator layout -> next conv's B fragments by v_permlane32_swap), no barrier other than
//     the weight ring's;
//   * every conv / gamma / beta matrix is carried as bf16 hi + lo and a product is hi*hi + hi*lo + lo*hi with
//     fp32 accumulation (16 significant bits per operand, ~1e-5 on the image -- there is no sine after these
//     GEMMs to amplify it, and bf16 keeps the fp32 exponent range so unnormalised activations cannot overflow);
//   * all weights of one tile form ONE linear stream in consumption order, pulled once per workgroup from L2 into
//     the LDS ring by LDS-DMA.
// Supported: C <= 256, per-pixel-style blocks only before the first skip block (true for map3d_mode mixed /
// isolated with the shipped mod_blocks); anything else stays on the fp32 engine (h3d_synthesis).
//
// X2 variant (template parameter, h3d_synthesis_x2): every conv / gamma / beta contraction in the "x2" arithmetic of
// x3_common.hpp -- f16 hi halves (11 bits; the reference trains these very convolutions under fp16 autocast,
// lib/trainers/base_trainer.py:50-51, so their inputs are f16-representable by construction) and ONE block-scaled fp6
// instruction for both cross terms.  Activations are not bounded here, so the fp6 record of a K-tile carries a per-lane
// (= per-pixel) power-of-two scale taken from the largest of the lane's 16 values.  The bilinear resize product keeps three
// bf16 products (12 small MFMAs per SPADE).
#include "x3_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

using namespace h3d;

namespace {

typedef BF16::vec8 bf8;
constexpr int kShared = 128;
#ifndef H3D_SYNTH_RING
#define H3D_SYNTH_RING 4
#endif
constexpr int kRingDepth = H3D_SYNTH_RING;
constexpr int kDescInts = 12;       // descriptor fields per block kept in LDS (multiple of 4: the ring behind them stays 16-byte aligned)
#ifndef H3D_SYNTH_VALU
#define H3D_SYNTH_VALU 5
#endif
constexpr int kValuPerMfma = H3D_SYNTH_VALU;
constexpr int kValuPerMfmaX2 = 0;  // x2: the scheduler places the producers' VALU work itself (24.27 -> 24.11 ms vs 5)    // VALU instructions slotted behind each MFMA of a section that carries epilogue work
#ifndef H3D_SYNTH_LOOK_X2
#define H3D_SYNTH_LOOK_X2 2
#endif
constexpr int kLook = 2;           // weight-fragment look-ahead in tile pairs (gemm_x3_roll)
template <int NT> constexpr int look_x2() { return H3D_SYNTH_LOOK_X2 < NT / 2 ? H3D_SYNTH_LOOK_X2 : NT / 2; }   // x2: sections are shorter      // 4 x 16 KB: leaves ~96 KB of LDS for the per-layer tables

struct Args {
    const unsigned char* stream;
    const float* tables;
    h3d_synth_desc D;
    const float* G;
    const float* cst;
    const float* ab;
    float* rgb;
    int table_floats, total_stages, g_channels, Hr, Wr, n_cst, n_ab, H, W, HdP, C, first_skip, n_pixel_blocks;
    float* state;          // [wave tiles][NT*4 + 1][64 lanes] float4: activations (+ rgb partial sums) between segments
    int load_state, store_state;
    int n_tiles;           // 128-pixel tiles per sample (a workgroup walks tiles blockIdx.x, + gridDim.x, ..)
    int n_walk, tile_first, tile_step;   // the tiles a launch covers: tile_first + i * tile_step, i < n_walk (all of them: 0, 1, n_tiles)
    int heads;             // x2 plans with ToRGB head tables: no zero table of "no ToRGB" weights in LDS (the producers carry no ToRGB)
    int mid_x3;            // x2 plans: the constant-style blocks in front of the first skip block travel / run in the x3 format (MIDX3)
    int* ovf;              // x2: int[B], ovf[b] set to 1 when an activation of sample b leaves the range the f16 planes carry (nullable)
    const int* run_if;     // int[B] (nullable): sample b is skipped when run_if[b] == 0 -- the guarded fallback of the x2 engine
};

// x2 range guard: the hi plane is f16 and the lo plane travels as f16(lo * 2^12) with |lo| <= ulp(hi) / 2, so both planes are
// finite exactly when |activation| < 2^15 (ulp 16 -> lo * 2^12 <= 2^15); beyond that the engine's result is not to be used.
constexpr float kX2ActLimit = 32768.f;

__device__ __forceinline__ float lrelu(float v) { return vmax(v, 0.2f * v); }

__device__ __forceinline__ float linspace_pm1(int n, int i) {
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// K order of every GEMM that consumes accumulators ("acc order", as in field_x3.hip): register r = 8j + e of accumulator
// tile t IS element e of the lane's B fragment of k-step 2t + j, so fragments are assembled without any cross-lane move;
// the host packs the K dimension of those weight matrices accordingly (pack_stream_bf16(acc_order=True)):
//     feature of k-slot (h, e) of k-step ks:  32*(ks/2) + (e & 3) + 8*(2*(ks & 1) + (e >> 2)) + 4*h
template <typename V8>
__device__ __forceinline__ void set_word(V8& f, int w, unsigned v) {
    u32x4 t = __builtin_bit_cast(u32x4, f);
    t[w] = v;
    f = __builtin_bit_cast(V8, t);
}

// fp32 values of one accumulator tile set -> bf16 hi/lo B fragments (all tiles at once; the progressive variant is
// SpadeProducer).  val(nt, rg) returns the 4 values of register group rg (rows 8rg + 4h + 0..3 of this lane's pixel).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// fp32 values of one accumulator tile set -> hi/lo B fragments (all tiles at once; the progressive variant is
// SpadeProducer).  val(nt, rg) returns the 4 values of register group rg (rows 8rg + 4h + 0..3 of this lane's pixel).
// X2: f16 hi fragments, lo' fragments and the fp6 record of every K-tile.
template <int NT, bool X2, typename V8, typename F>
__device__ __forceinline__ void make_frags(V8 (&xh)[2 * NT], V8 (&xl)[2 * NT], i32x8 (&b6)[NT], f32x16 (&src)[NT], float& gmax, F val) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        pin1(src[nt]);                       // the source tile sits in AGPRs until this iteration reads it
        float amax = 0.f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 y = val(nt, rg);
            unsigned l0, l1, h0, h1;
            if constexpr (X2) {
                amax = vmax3_abs2(vmax3_abs2(amax, y.x, y.y), y.z, y.w);
                h0 = split2_x2(y.x, y.y, l0); h1 = split2_x2(y.z, y.w, l1);
            } else {
                h0 = split2_bf16(y.x, y.y, l0); h1 = split2_bf16(y.z, y.w, l1);
            }
            set_word(xh[2 * nt + (rg >> 1)], 2 * (rg & 1) + 0, h0);
            set_word(xh[2 * nt + (rg >> 1)], 2 * (rg & 1) + 1, h1);
            set_word(xl[2 * nt + (rg >> 1)], 2 * (rg & 1) + 0, l0);
            set_word(xl[2 * nt + (rg >> 1)], 2 * (rg & 1) + 1, l1);
        }
        if constexpr (X2) {
            b6[nt] = x2_record_dyn(xl[2 * nt], xl[2 * nt + 1], xh[2 * nt], xh[2 * nt + 1], amax);
            gmax = vmax(gmax, amax);
            asm volatile("" : "+v"(b6[nt]), "+v"(xh[2 * nt]), "+v"(xh[2 * nt + 1]));
        } else {
            asm volatile("" : "+v"(xl[2 * nt]), "+v"(xl[2 * nt + 1]), "+v"(xh[2 * nt]), "+v"(xh[2 * nt + 1]));
        }
        // The launder above anchors this tile's arithmetic HERE: instruction selection schedules for register pressure and is
        // free to sink pure arithmetic below later loads (nothing but data flow orders it against the sched_barrier), which
        // with immediate-offset table reads meant: all 64 table reads of the eight tiles first (256 live registers, spilled),
        // every tile's arithmetic afterwards.
        // bound the scheduler's load hoisting to one tile: the table reads of all tiles at once would cost
        // hundreds of registers on top of the resident activations
        __builtin_amdgcn_sched_barrier(0);
    }
}

// two fp32 -> packed bf16 hi halves (returned) and packed bf16 lo halves; plain VALU only (VOP3P instructions beside
// MFMAs cost several issue slots on this chip, and hipcc's SLP vectoriser would pack the two subtractions).
__device__ __forceinline__ unsigned split2_plain(float a, float b, unsigned& lo) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, b2));
    const float fa = __builtin_bit_cast(float, hb << 16), fb = __builtin_bit_cast(float, hb & 0xffff0000u);
    float la, lb;
    asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(fa));
    asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(fb));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{la, lb}, b2));
    return hb;
}

// lrelu(t) from u = 0.4 t (the constant-style affine tables arrive pre-multiplied): 0.6 t + 0.4 |t| = 1.5 u + |u|, one instruction
__device__ __forceinline__ float lrelu_from_scaled(float u) {
    float z;
    asm("v_fma_f32 %0, %1, %2, |%1|" : "=v"(z) : "v"(u), "s"(1.5f));
    return z;
}

__device__ __forceinline__ float lrelu_plain(float v) {
    float t;
    asm("v_mul_f32 %0, 0x3e4ccccd, %1" : "=v"(t) : "v"(v));      // 0.2 * v, kept out of the SLP vectoriser's reach
    return vmax(v, t);
}

// Producer of a conv's B fragments from a feature-major accumulator set, one 32-channel tile in eight chunks of two
// activations (so that the consuming GEMM hides one chunk behind each tile-pair section of k-steps 2t, 2t+1):
//   AFFINE  y = lrelu(v * sc + sh)   tables [HdP/2][4] = 0.4 * (sc[n], sc[n+1], sh[n], sh[n+1]) (constant-style SPADE with the
//           conv bias of v folded into sh; the 0.4: lrelu_from_scaled);  otherwise y = lrelu(v) (per-pixel style: v is already modulated)
//   RGB     also accumulates the ToRGB of v (the previous block's output) into rgb[3]: wr = LDS [3][HdP]
// Source values are read from the AGPRs eight at a time: a v_accvgpr_read between MFMAs waits for the matrix pipe.
template <int NT, bool AFFINE, bool RGB, bool X2, typename V8>
struct SpadeProducer {
    f32x16 (&src)[NT];
    V8 (&xh)[2 * NT];
    V8 (&xl)[2 * NT];
    i32x8 (&b6)[NT];
    lds_ptr ab;               // lane base (+ 32 h bytes) of the [HdP/2][4] affine table
    lds_ptr wr;               // lane base (+ 16 h bytes) of the [3][HdP] ToRGB weights
    float (&rgb)[3];
    float& gmax;              // x2: largest |activation| this lane has fed to the matrix cores (range guard)
    f32x4 tv;
    f32x2 w0, w1, w2;
    float sv[8];
    float amax;               // x2: largest |activation| of the tile in production (this lane)

    template <int TILE, int C>
    static constexpr int chan() { return TILE * 32 + (C / 2) * 8 + (C % 2) * 2; }      // + 4 h: in the lane bases
    template <int TILE, int C>
    __device__ __forceinline__ void fetch() {
        constexpr int n = chan<TILE, C>(), HdP = NT * 32;
        if constexpr (AFFINE) tv = ldt4(ab, 2 * n);
        if constexpr (RGB) {
            w0 = lds_ld<f32x2>(wr + 4 * n);
            w1 = lds_ld<f32x2>(wr + 4 * (HdP + n));
            w2 = lds_ld<f32x2>(wr + 4 * (2 * HdP + n));
        }
    }
    __device__ __forceinline__ void prime() { fetch<0, 0>(); }
    template <int TILE, int C>
    __device__ __forceinline__ void chunk() {
        if constexpr (C == 0) pin1(src[TILE]);
        constexpr int rg = C / 2;
        if constexpr (C % 4 == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[i] = src[TILE][(C / 4) * 8 + i];
        }
        const float s0 = sv[(C % 4) * 2], s1 = sv[(C % 4) * 2 + 1];
        float y0 = s0, y1 = s1;
        if constexpr (AFFINE) { y0 = fmaf(s0, tv.x, tv.z); y1 = fmaf(s1, tv.y, tv.w); }
        if constexpr (RGB) {
            rgb[0] = fmaf(s1, w0.y, fmaf(s0, w0.x, rgb[0]));
            rgb[1] = fmaf(s1, w1.y, fmaf(s0, w1.x, rgb[1]));
            rgb[2] = fmaf(s1, w2.y, fmaf(s0, w2.x, rgb[2]));
        }
        if constexpr (C < 7) fetch<TILE, C + 1>();
        else if constexpr (TILE + 1 < NT) fetch<TILE + 1, 0>();
        unsigned lo, hi;
        const float z0 = AFFINE ? lrelu_from_scaled(y0) : lrelu_plain(y0), z1 = AFFINE ? lrelu_from_scaled(y1) : lrelu_plain(y1);
        if constexpr (X2) {
            amax = C == 0 ? vmax_abs2(z0, z1) : vmax3_abs2(amax, z0, z1);
            hi = split2_x2(z0, z1, lo);
        } else {
            hi = split2_plain(z0, z1, lo);
        }
        // registers 4*rg + 2*(C%2) + {0, 1} of the tile = k-step 2*TILE + (rg >> 1), word 2*(rg & 1) + C%2
        set_word(xh[2 * TILE + (rg >> 1)], 2 * (rg & 1) + (C % 2), hi);
        set_word(xl[2 * TILE + (rg >> 1)], 2 * (rg & 1) + (C % 2), lo);
    }
    // x2: fp6 record of the finished tile (called before chunk<TILE + 1, 0> restarts the running maximum)
    template <int TILE>
    __device__ __forceinline__ void convert() {
        if constexpr (X2) {
            b6[TILE] = x2_record_dyn(xl[2 * TILE], xl[2 * TILE + 1], xh[2 * TILE], xh[2 * TILE + 1], amax);
            gmax = vmax(gmax, amax);
        }
    }
};

// conv GEMM dst (+)= W * frags(producer(src)) with the fragment epilogue of tile t+1 hidden behind the MFMAs of k-steps
// 2t, 2t+1 (which only need tile t); only tile 0's epilogue is exposed.  src and dst are different register sets.
// HEAD (x2 only, round 5): the ToRGB of everything downstream of this convolution rides along as a NINTH output tile on
// pre-multiplied weights -- rgb = sum_k Wr_k x_k over the skip blocks is (sum_k Wr_k) x_base + sum_j [(sum_{k >= j} Wr_k) W1_j] y_j,
// so block j's second convolution also multiplies its input fragments y_j with the 3 x C matrix M_j = V_j W1_j (host:
// SynthesisPlan.build_x3) in the same x2 arithmetic: per k-step one f16 product, per K-tile one fp6 product, accumulated in `hacc`
// across ALL skip blocks.  Its A operand comes from a 4 KB LDS table per block -- [k-step][f16 hi fragment | half of the fp6
// record][4 rows x 2 lane halves][16 B], lane (m, h) reads row m & 3 -- one k-step ahead of its use.  Replaces the 3 FMAs per
// activation that the ToRGB used to cost in the producer of the NEXT block's first convolution (1 920 vector instructions per tile).
template <int NT, bool ZERO, bool X2, bool HEAD = false, typename V8, typename RING, typename PROD>
__device__ __forceinline__ void conv_progressive(f32x16 (&dst)[NT], V8 (&xh)[2 * NT], V8 (&xl)[2 * NT], i32x8 (&b6)[NT], RING& ring, PROD& prod,
                                                 f32x16* hacc = nullptr, lds_ptr hbase = nullptr) {
    static_assert(!HEAD || X2, "the ToRGB head tile exists in the x2 arithmetic only");
    u32x4 hfrag = {}, hr0 = {}, hr1 = {};
    (void)hfrag; (void)hr0; (void)hr1;
    prod.prime();
    static_for<0, 8>([&](auto c) __attribute__((always_inline)) { prod.template chunk<0, decltype(c)::value>(); });
    __builtin_amdgcn_sched_barrier(0);
    H3D_TRACE(30);
    constexpr int W = NT, PER = 8 / W;          // sections per 2-k-step window, chunks per section
    auto hook = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int t = g / W + 1, j = g % W;
        if constexpr (X2 && j == 0 && t - 1 < NT) prod.template convert<t - 1>();       // first section of k-step 2 (t - 1)
        if constexpr (t < NT) {
            static_for<0, PER>([&](auto q) __attribute__((always_inline)) { prod.template chunk<t, j * PER + decltype(q)::value>(); });
        }
        if constexpr (HEAD) {
            constexpr int P = NT / 2, ks = g / P, p = g % P;
            if constexpr (p == 0) {                   // this k-step's head fragment (and, in an odd k-step, the K-tile's record halves)
                hfrag = lds_ld<u32x4>(hbase + ks * 256);
                if constexpr (ks % 2 == 1) {
                    hr0 = lds_ld<u32x4>(hbase + (ks - 1) * 256 + 128);
                    hr1 = lds_ld<u32x4>(hbase + ks * 256 + 128);
                }
            }
            if constexpr (p == P / 2) {               // two sections later: the reads have landed behind the section's own matrix work
                *hacc = F16::mfma(__builtin_bit_cast(F16::vec8, hfrag), xh[ks], *hacc);
                if constexpr (ks % 2 == 1) {
                    const i32x8 w6 = {(int)hr0[0], (int)hr0[1], (int)hr0[2], (int)hr0[3], (int)hr1[0], (int)hr1[1], (int)hr1[2], 0};
                    *hacc = mm6<false>(w6, b6[ks / 2], *hacc);
                }
            }
        }
    };
    if constexpr (X2) {
        const F16::vec8 none[1] = {};
        gemm_x2_roll<NT, 2 * NT, 0, 2 * NT, NT, false, look_x2<NT>(), kValuPerMfmaX2, ZERO>(dst, xh, b6, none, ring, hook);
    } else {
        gemm_x3_roll<BF16, NT, 2 * NT, 2 * NT, false, kLook, kValuPerMfma, ZERO>(dst, xh, xl, ring, hook);
    }
}

// MIDX3 (x2 plans, round 6): the constant-style blocks between the per-pixel blocks and the first skip block (block 3 of the shipped
// configs: the base of the residual stream, whose error every later block and every ToRGB head inherits -- the largest single
// contribution in the per-contraction attribution, profiles/r5_x2_error_attribution_item14.txt) run on three bf16 products: their
// stages travel in the x3 format and their fragments are bf16 hi / lo.  Same registers (the 64 of the fp6 records hold the lo
// fragments), same ring (single-stage acquires, as the x3 tail of gemm_x2_roll).
template <int NT, int DEPTH, bool SEG, bool X2, bool HEADS = false, bool MIDX3 = false>
__global__ __launch_bounds__(256, 1) void synthesis_x3_kernel(Args A) {
    constexpr int KS = 2 * NT, kHdP = NT * 32;       // the host sets A.HdP = 32 NT: table strides are compile-time
    constexpr bool kHeads = HEADS;                   // ToRGB of the skip blocks as a ninth tile of their second convolution (conv_progressive)
    static_assert(!HEADS || (X2 && !SEG), "ToRGB heads: single-launch x2 plans only");
    static_assert(!MIDX3 || (X2 && !SEG), "three-product middle blocks: single-launch x2 plans only");
    typedef typename std::conditional<X2, F16, BF16>::type T;
    typedef typename T::vec8 frag8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HdP = A.HdP;
    float* tab0 = smem;                                      // [table_floats] static tables (descriptor offsets)
    float* ab0 = tab0 + ((A.table_floats + 3) & ~3);         // [n_ab][2][HdP] this sample's constant-style affines
    float* cst0 = ab0 + A.n_ab * 2 * HdP;                    // [n_cst][128]   this sample's shared-MLP constants
    float* zero0 = cst0 + A.n_cst * kShared;                 // [3][HdP] zeros: ToRGB weights of "no ToRGB"
    int* dtab = reinterpret_cast<int*>(zero0 + (HEADS ? 0 : 3 * HdP));     // [H3D_MAX_BLOCKS + 1][kDescInts] the descriptor fields the block loops read
    unsigned char* ring_lds = reinterpret_cast<unsigned char*>(dtab + (H3D_MAX_BLOCKS + 1) * kDescInts);

    if (A.run_if && A.run_if[blockIdx.y] == 0) return;          // guarded fallback: nothing to redo for this sample
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int64_t HW = (int64_t)A.H * A.W;
    float gmax = 0.f;                                // x2: running maximum of |activation| over everything this lane converts
    (void)gmax;
    // The descriptor is a kernel argument: it lives in the (host-visible) kernarg segment, and a scalar load from there that misses
    // the scalar cache deep inside the kernel was measured at ~50 000 cycles (cycle trace: 57 % of every skip block went into the
    // two loads of "to_rgb / w_rgb" behind the second convolution).  All lanes fetch their block's fields once, here, in parallel;
    // the loops read the copy in LDS (dget: wave-uniform).
    if (t <= H3D_MAX_BLOCKS) {
        int* d = dtab + t * kDescInts;
        if (t < H3D_MAX_BLOCKS) {
            const h3d_block_desc& bd = A.D.block[t];
            d[0] = bd.to_rgb; d[1] = (int)bd.w_rgb; d[2] = bd.skip;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                d[3 + 4 * q] = bd.spade[q].ab_index; d[4 + 4 * q] = bd.spade[q].cst_index;
                d[5 + 4 * q] = bd.spade[q].g_offset; d[6 + 4 * q] = (int)bd.spade[q].vec;
            }
            d[11] = (int)bd.spade[1].b_conv;         // skip blocks of an x2 plan with ToRGB heads: float offset of the block's head table (else -1)
        } else {
            d[0] = A.D.n_blocks; d[1] = (int)A.D.w_in; d[2] = (int)A.D.b_in;
        }
    }
    auto dget = [&](int blk, int k) { return __builtin_amdgcn_readfirstlane(dtab[blk * kDescInts + k]); };

    for (int i = t; i < A.table_floats; i += 256) tab0[i] = A.tables[i];
    for (int i = t; i < A.n_ab * 2 * HdP; i += 256) ab0[i] = A.ab[(int64_t)b * A.n_ab * 2 * HdP + i];
    for (int i = t; i < A.n_cst * kShared; i += 256) cst0[i] = A.cst[(int64_t)b * A.n_cst * kShared + i];
    if constexpr (!HEADS) {
        for (int i = t; i < 3 * HdP; i += 256) zero0[i] = 0.f;
    }
    __syncthreads();

    H3D_TRACE_INIT();
    H3D_TRACE(0);
    WeightRing<NT, DEPTH + (X2 ? 1 : 0), X2 ? 1 : 0> ring;        // x2: the fp6 records span two stages (LAG = 1)
    ring.init(A.stream, ring_lds, A.total_stages, wave, lane);

    // ---- this lane's pixel: synthesis-input coordinates and bilinear taps into the low-res maps
    const int n_blocks = dget(H3D_MAX_BLOCKS, 0);
    // kernel arguments used late in the kernel, pinned in scalar registers now (the compiler would otherwise re-load them from the
    // kernarg segment where they are used -- see above)
    int first_skip = A.first_skip, n_pixel_blocks = A.n_pixel_blocks, n_tiles = A.n_tiles;
    // the tiles of this workgroup: tile_first + (blockIdx.x + i * gridDim.x) * tile_step < n_tiles, as ONE running index
    int tile_stride = (int)gridDim.x * A.tile_step;
    float* rgb_out = A.rgb;
    asm volatile("" : "+s"(first_skip), "+s"(n_pixel_blocks), "+s"(rgb_out), "+s"(n_tiles), "+s"(tile_stride));
    // Persistent workgroups: a workgroup walks the 128-pixel tiles blockIdx.x, blockIdx.x + gridDim.x, .. of its sample with the
    // tables above staged once and the weight ring running across tile boundaries (the stream wraps exactly at the end of the
    // network, so the first stages of the next tile are prefetched under the last layers of this one).
#pragma unroll 1
    for (int tile = A.tile_first + (int)blockIdx.x * A.tile_step; tile < n_tiles; tile += tile_stride) {
    const int64_t p_tile = ((int64_t)tile * 4 + wave) * 32;
    int64_t p = p_tile + m;
    const bool okp = p < HW;
    if (!okp) p = min(p_tile, HW - 1);          // lanes past the image shadow the tile's first pixel (never stored)
    const int Y = (int)(p / A.W), X = (int)(p % A.W);
    const float ci = linspace_pm1(A.H, Y), cj = linspace_pm1(A.W, X);
    float sy = fmaxf(((float)Y + 0.5f) * ((float)A.Hr / (float)A.H) - 0.5f, 0.f);
    float sx = fmaxf(((float)X + 0.5f) * ((float)A.Wr / (float)A.W) - 0.5f, 0.f);
    const int y0 = min((int)sy, A.Hr - 1), x0 = min((int)sx, A.Wr - 1);
    const float ty = sy - (float)y0, tx = sx - (float)x0;
    const float* Gb_ = A.G + (int64_t)b * A.Hr * A.Wr * A.g_channels;
    asm volatile("" : "+s"(Gb_));                  // loaded from the kernarg segment here, not where it is first used
    const float* __restrict__ Gb = Gb_;
    // Bilinear resize of the low-res maps as a matrix product (the usual case: the wave's 32 pixels lie in one image
    // row and touch at most 8 low-res columns): D[ch][px] = sum_k T[k][ch] * wi[k][px] over the 16 texels
    // k = 8*r + e <-> (row ya + r, column xa + e), wi = the pixel's bilinear weights (<= 4 non-zeros).  Edge clamping
    // happens in the texel addresses, so a clamped x1 / y1 lands on the same texel with the summed weight.
    const int ya = __builtin_amdgcn_readfirstlane(y0), xa = __builtin_amdgcn_readfirstlane(x0);
    const int Ya = __builtin_amdgcn_readfirstlane(Y);
    const int jx = x0 - xa;
    // the host guarantees this geometry (h3d_synthesis_x3 refuses anything else); a violation must not pass silently
    if (A.g_channels > 0 && !__all(Y == Ya && jx >= 0 && jx <= 6)) __builtin_trap();
    bf8 wih, wil;
    {
        const float wy = h ? ty : 1.f - ty;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float w0 = wy * (e == jx ? 1.f - tx : (e == jx + 1 ? tx : 0.f));
            const float w1 = wy * (e + 1 == jx ? 1.f - tx : (e == jx ? tx : 0.f));
            hw[e / 2] = split2_bf16(w0, w1, lw[e / 2]);
        }
        wih = __builtin_bit_cast(bf8, u32x4{hw[0], hw[1], hw[2], hw[3]});
        wil = __builtin_bit_cast(bf8, u32x4{lw[0], lw[1], lw[2], lw[3]});
    }
    // texel row of this half-wave (k = 8h + e) and the per-column offsets (wave-uniform)
    // explicitly a GLOBAL pointer: laundered through the asm above the address space is unknown to the compiler, and a generic
    // pointer makes these FLAT loads -- while a FLAT access is pending the waitcnt pass turns every later LDS wait into lgkmcnt(0)
    typedef const __attribute__((address_space(1))) float* gptr;
    const gptr trow = (gptr)(Gb + ((int64_t)min(ya + h, A.Hr - 1) * A.Wr) * A.g_channels + m);
    f32x16 x[NT];
    frag8 xh[KS], xl[KS];
    i32x8 b6[NT];
    float rgb_acc[3] = {0.f, 0.f, 0.f};
    f32x16 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // kHeads: rows r = head r & 3, all skip blocks
    (void)hacc;

    const int64_t wtile = (int64_t)b * n_tiles * 4 + (int64_t)tile * 4 + wave;
    float4* st_io = reinterpret_cast<float4*>(A.state) + wtile * (NT * 4 + 1) * 64 + lane;
    if (SEG && A.load_state) {
        // ---- resume: activations (lane-linear, as the previous segment left them) and the ToRGB partial sums
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 v = st_io[(nt * 4 + rg) * 64];
                x[nt][rg * 4 + 0] = v.x; x[nt][rg * 4 + 1] = v.y; x[nt][rg * 4 + 2] = v.z; x[nt][rg * 4 + 3] = v.w;
            }
        const float4 r = st_io[NT * 4 * 64];
        rgb_acc[0] = r.x; rgb_acc[1] = r.y; rgb_acc[2] = r.z;
    } else {
        // ---- A8: x0 = sin(w0*i + w1*j + b) in accumulator layout
        const lds_ptr win = lane_base(tab0 + dget(H3D_MAX_BLOCKS, 1), 16 * h);
        const lds_ptr bin = lane_base(tab0 + dget(H3D_MAX_BLOCKS, 2), 16 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8;
                const f32x4 w0 = ldt4(win, n);
                const f32x4 w1 = ldt4(win, kHdP + n);
                const f32x4 bb = ldt4(bin, n);
                x[nt][rg * 4 + 0] = sin_hw(w0.x * ci + w1.x * cj + bb.x);
                x[nt][rg * 4 + 1] = sin_hw(w0.y * ci + w1.y * cj + bb.y);
                x[nt][rg * 4 + 2] = sin_hw(w0.z * ci + w1.z * cj + bb.z);
                x[nt][rg * 4 + 3] = sin_hw(w0.w * ci + w1.w * cj + bb.w);
            }
            pin1(x[nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // constant-style SPADE: y = lrelu(v * a + b) -> fragments
    auto const_frags = [&](f32x16 (&v)[NT], const float* ab_) {      // ab: [HdP/2][4] = sc[n], sc[n+1], sh[n], sh[n+1]
        const lds_ptr ab = lane_base(ab_, 32 * h);
        make_frags<NT, X2>(xh, xl, b6, v, gmax, [&](int nt, int rg) {
            const int n = nt * 32 + rg * 8;
            const f32x4 ta = ldt4(ab, 2 * n), tb = ldt4(ab, 2 * n + 4);
            float4 y;
            y.x = lrelu_from_scaled(fmaf(v[nt][rg * 4 + 0], ta.x, ta.z));
            y.y = lrelu_from_scaled(fmaf(v[nt][rg * 4 + 1], ta.y, ta.w));
            y.z = lrelu_from_scaled(fmaf(v[nt][rg * 4 + 2], tb.x, tb.z));
            y.w = lrelu_from_scaled(fmaf(v[nt][rg * 4 + 3], tb.y, tb.w));
            return y;
        });
    };
    // accumulator initialisation with a per-channel vector (1 + gamma bias of the per-pixel SPADE)
    auto set_bias = [&](f32x16 (&v)[NT], lds_ptr bc) {               // bc: lane base (+ 16 h bytes)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 bb = ldt4(bc, nt * 32 + rg * 8);
                v[nt][rg * 4 + 0] = bb.x; v[nt][rg * 4 + 1] = bb.y; v[nt][rg * 4 + 2] = bb.z; v[nt][rg * 4 + 3] = bb.w;
            }
            pin1(v[nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ToRGB: rgb += Wrgb * x + b, an fp32 dot product over this lane's half of the channels
    auto to_rgb = [&](const float* wr_, bool with_bias) {
        const lds_ptr wr = lane_base(wr_, 16 * h);
        constexpr int HdP = kHdP;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            pin1(x[nt]);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8;
                const f32x4 w0 = ldt4(wr, n);
                const f32x4 w1 = ldt4(wr, HdP + n);
                const f32x4 w2 = ldt4(wr, 2 * HdP + n);
                const float v0 = x[nt][rg * 4 + 0], v1 = x[nt][rg * 4 + 1], v2 = x[nt][rg * 4 + 2], v3 = x[nt][rg * 4 + 3];
                s0 = fmaf(v3, w0.w, fmaf(v2, w0.z, fmaf(v1, w0.y, fmaf(v0, w0.x, s0))));
                s1 = fmaf(v3, w1.w, fmaf(v2, w1.z, fmaf(v1, w1.y, fmaf(v0, w1.x, s1))));
                s2 = fmaf(v3, w2.w, fmaf(v2, w2.z, fmaf(v1, w2.y, fmaf(v0, w2.x, s2))));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 0 && with_bias) { s0 += wr_[3 * HdP + 0]; s1 += wr_[3 * HdP + 1]; s2 += wr_[3 * HdP + 2]; }
        rgb_acc[0] += s0; rgb_acc[1] += s1; rgb_acc[2] += s2;
    };

    // Register roles: every SPADE+conv stage takes its input in x and leaves its output in x (the conv accumulates
    // onto x = bias); acc is the scratch accumulator of the gamma / beta GEMMs and of the first conv of a skip block.
    // ================= blocks before the first skip connection ===================================================
    // Per-pixel-style blocks first, then constant-style ones, in two loops without a per-SPADE branch (the host checks
    // that the styles are laid out that way, as mod_blocks = [0, 1, 2] is): at the control-flow merge of the two kinds of
    // SPADE hipcc could not keep both accumulator sets in the 256 AGPRs and spilled three tiles (48 registers, 5 GB of
    // scratch stores per launch at 512^2 x 16) at every SPADE.
#pragma unroll 1
    for (int blk = 0; blk < n_pixel_blocks; ++blk) {
        int opaque = 0;                       // keeps the loop-invariant LDS table loads inside the body
        asm volatile("" : "+s"(opaque));
        const float* tab = tab0 + opaque;
        const float* abt = ab0 + opaque;
        const float* cstt = cst0 + opaque;
        (void)abt;
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            f32x16 acc[NT];
            // ---- shared-MLP activations a = relu(resize(G) + cst) of this lane's pixel as B fragments
            frag8 ah[8], al[8];
            i32x8 a6[4];
            const float* cs = cstt + dget(blk, 4 + 4 * s) * kShared;
            const int g_off = dget(blk, 5 + 4 * s);
            {
                float tq[4][8];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        tq[tt][e] = trow[(int64_t)min(xa + e, A.Wr - 1) * A.g_channels + g_off + 32 * tt];
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                f32x16 (&d)[4] = reinterpret_cast<f32x16(&)[4]>(acc);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) hw[e / 2] = split2_bf16(tq[tt][e], tq[tt][e + 1], lw[e / 2]);
                    const bf8 th = __builtin_bit_cast(bf8, u32x4{hw[0], hw[1], hw[2], hw[3]});
                    const bf8 tl = __builtin_bit_cast(bf8, u32x4{lw[0], lw[1], lw[2], lw[3]});
                    d[tt] = BF16::mfma(th, wih, zero);
                    d[tt] = BF16::mfma(th, wil, d[tt]);
                    d[tt] = BF16::mfma(tl, wih, d[tt]);
                }
                const lds_ptr csl = lane_base(cs, 16 * h);
                make_frags<4, X2>(ah, al, a6, d, gmax, [&](int nt, int rg) {
                    const f32x4 k4 = ldt4(csl, nt * 32 + rg * 8);
                    float4 y;
                    y.x = vrelu(d[nt][rg * 4 + 0] + k4.x);
                    y.y = vrelu(d[nt][rg * 4 + 1] + k4.y);
                    y.z = vrelu(d[nt][rg * 4 + 2] + k4.z);
                    y.w = vrelu(d[nt][rg * 4 + 3] + k4.w);
                    return y;
                });
            }
            const lds_ptr vec = lane_base(tab + dget(blk, 6 + 4 * s), 16 * h);
            // gamma:  acc = 1 + gamma ;  x <- (x*sc + sh) * acc + beta_bias   (beta accumulates on top of x)
            set_bias(acc, vec);
            pin_agpr<NT>(x); pin_agpr<NT>(acc);
            const F16::vec8 none[1] = {};
            (void)none;
            if constexpr (X2) gemm_x2_roll<NT, 8, 0, 8, 4, false, look_x2<NT>()>(acc, ah, a6, none, ring);
            else gemm_x3_roll<BF16, NT, 8, 8, false, kLook>(acc, ah, al, ring);
            pin_agpr<NT>(x); pin_agpr<NT>(acc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                pin1(x[nt]); pin1(acc[nt]);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = nt * 32 + rg * 8;
                    const f32x4 bt = ldt4(vec, kHdP + n);
                    const f32x4 sc = ldt4(vec, 2 * kHdP + n);
                    const f32x4 sh = ldt4(vec, 3 * kHdP + n);
                    acc[nt][rg * 4 + 0] = fmaf(fmaf(x[nt][rg * 4 + 0], sc.x, sh.x), acc[nt][rg * 4 + 0], bt.x);
                    acc[nt][rg * 4 + 1] = fmaf(fmaf(x[nt][rg * 4 + 1], sc.y, sh.y), acc[nt][rg * 4 + 1], bt.y);
                    acc[nt][rg * 4 + 2] = fmaf(fmaf(x[nt][rg * 4 + 2], sc.z, sh.z), acc[nt][rg * 4 + 2], bt.z);
                    acc[nt][rg * 4 + 3] = fmaf(fmaf(x[nt][rg * 4 + 3], sc.w, sh.w), acc[nt][rg * 4 + 3], bt.w);
                }
                pin1(acc[nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // beta:   y = lrelu(acc + beta)
            pin_agpr<NT>(acc);
            if constexpr (X2) gemm_x2_roll<NT, 8, 0, 8, 4, false, look_x2<NT>()>(acc, ah, a6, none, ring);
            else gemm_x3_roll<BF16, NT, 8, 8, false, kLook>(acc, ah, al, ring);
            pin_agpr<NT>(acc);
            {
                SpadeProducer<NT, false, false, X2, frag8> prod{acc, xh, xl, b6, nullptr, nullptr, rgb_acc, gmax};
                conv_progressive<NT, true, X2>(x, xh, xl, b6, ring, prod);
            }
            pin_agpr<NT>(x);
        }
        if (dget(blk, 0)) to_rgb(tab + dget(blk, 1), true);
    }
#pragma unroll 1
    for (int blk = n_pixel_blocks; blk < first_skip; ++blk) {
        int opaque = 0;                       // keeps the loop-invariant LDS table loads inside the body
        asm volatile("" : "+s"(opaque));
        const float* tab = tab0 + opaque;
        const float* abt = ab0 + opaque;
        const float* cstt = cst0 + opaque;
        (void)cstt;
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            // constant style before the first skip block: x is both source and destination, so the fragments
            // are completed before the conv starts
            if constexpr (X2 && MIDX3) {
                // three bf16 products for this convolution (see MIDX3 above): bf16 hi / lo fragments of lrelu(x a + b), the x3 GEMM
                // on the x2 kernel's ring; the same table as the x2 path (the 0.4-scaled affine)
                typedef BF16::vec8 bfrag;
                bfrag yh[KS], yl[KS];
                const lds_ptr ab = lane_base(abt + dget(blk, 3 + 4 * s) * 2 * HdP, 32 * h);
                make_frags<NT, false>(yh, yl, b6, x, gmax, [&](int nt, int rg) {
                    const int n = nt * 32 + rg * 8;
                    const f32x4 ta = ldt4(ab, 2 * n), tb = ldt4(ab, 2 * n + 4);
                    float4 y;
                    y.x = lrelu_from_scaled(fmaf(x[nt][rg * 4 + 0], ta.x, ta.z));
                    y.y = lrelu_from_scaled(fmaf(x[nt][rg * 4 + 1], ta.y, ta.w));
                    y.z = lrelu_from_scaled(fmaf(x[nt][rg * 4 + 2], tb.x, tb.z));
                    y.w = lrelu_from_scaled(fmaf(x[nt][rg * 4 + 3], tb.y, tb.w));
                    return y;
                });
                gemm_x3_roll<BF16, NT, KS, KS, false, kLook, 0, true>(x, yh, yl, ring);
            } else {
                const_frags(x, abt + dget(blk, 3 + 4 * s) * 2 * HdP);
                if constexpr (X2) {
                    const F16::vec8 none[1] = {};
                    gemm_x2_roll<NT, KS, 0, KS, NT, false, look_x2<NT>(), 0, true>(x, xh, b6, none, ring);
                } else {
                    gemm_x3_roll<BF16, NT, KS, KS, false, kLook, 0, true>(x, xh, xl, ring);
                }
            }
            pin_agpr<NT>(x);
        }
        if (dget(blk, 0)) to_rgb(tab + dget(blk, 1), true);
    }

    // ================= blocks from the first skip connection on (constant style only) ============================
    // conv 0 goes x -> acc (x stays live as the residual), conv 1 goes acc -> x accumulating onto the residual
    // (conv biases are folded into the consumers' tables by the host: build_x3 in synthesis_pack.py).
    H3D_TRACE_RESET();
#pragma unroll 1
    for (int blk = first_skip; blk < n_blocks; ++blk) {
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* tab = tab0 + opaque;
        const float* abt = ab0 + opaque;
        f32x16 acc[NT];
        pin_agpr<NT>(x);
        H3D_TRACE(20);
        {   // conv 0, with the ToRGB of the previous skip block's output (this block's input x) riding along
            const float* wr_prev = (blk > first_skip && dget(blk - 1, 0)) ? tab + dget(blk - 1, 1) : zero0 + opaque;
#ifdef H3D_EXPERIMENT_NO_RGB
            SpadeProducer<NT, true, false, X2, frag8> prod{x, xh, xl, b6, lane_base(abt + dget(blk, 3) * 2 * HdP, 32 * h), lane_base(wr_prev, 16 * h), rgb_acc, gmax};
#else
            SpadeProducer<NT, true, !kHeads, X2, frag8> prod{x, xh, xl, b6, lane_base(abt + dget(blk, 3) * 2 * HdP, 32 * h), lane_base(wr_prev, 16 * h), rgb_acc, gmax};
#endif
            conv_progressive<NT, true, X2>(acc, xh, xl, b6, ring, prod);
        }
        pin_agpr<NT>(x); pin_agpr<NT>(acc);
        H3D_TRACE(21);
        {
            SpadeProducer<NT, true, false, X2, frag8> prod{acc, xh, xl, b6, lane_base(abt + dget(blk, 7) * 2 * HdP, 32 * h), nullptr, rgb_acc, gmax};
            if constexpr (kHeads) {
                conv_progressive<NT, false, X2, true>(x, xh, xl, b6, ring, prod, &hacc, lane_base(tab + dget(blk, 11), ((lane & 3) + 4 * h) * 16));
            } else {
                conv_progressive<NT, false, X2>(x, xh, xl, b6, ring, prod);
            }
        }
        pin_agpr<NT>(x);
        if (dget(blk, 0) && h == 0) {          // bias of this block's ToRGB; its weights ride in the next block's conv 0
            const float* wb = tab + dget(blk, 1) + 3 * HdP;
            rgb_acc[0] += wb[0]; rgb_acc[1] += wb[1]; rgb_acc[2] += wb[2];
        }
    }
    H3D_TRACE(5);
    if (n_blocks > first_skip && dget(n_blocks - 1, 0)) to_rgb(tab0 + dget(n_blocks - 1, 1), false);
    if constexpr (kHeads) {
        // the head tile: every lane holds the complete sums of its pixel (rows = heads, replicated); the store below adds the two
        // lane halves, so only one of them contributes
        if (h == 0) { rgb_acc[0] += hacc[0]; rgb_acc[1] += hacc[1]; rgb_acc[2] += hacc[2]; }
    }
    H3D_TRACE(6);
    if (SEG && A.store_state) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                st_io[(nt * 4 + rg) * 64] = make_float4(x[nt][rg * 4 + 0], x[nt][rg * 4 + 1], x[nt][rg * 4 + 2], x[nt][rg * 4 + 3]);
        st_io[NT * 4 * 64] = make_float4(rgb_acc[0], rgb_acc[1], rgb_acc[2], 0.f);
        continue;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = rgb_acc[c] + __shfl_xor(rgb_acc[c], 32, 64);
        // a GLOBAL store (see trow above): as a generic pointer this is a FLAT store, and a FLAT access that is never waited for
        // (stores are not) keeps "pending flat" alive around the persistent tile loop: EVERY LDS wait of the kernel then became
        // s_waitcnt lgkmcnt(0) -- the weight-fragment look-ahead was drained at every table read (round 5, found in the ISA)
        if (okp && h == 0) ((__attribute__((address_space(1))) float*)rgb_out)[((int64_t)b * 3 + c) * HW + p] = v;
    }
    }   // tiles
    if constexpr (X2) {
        // sticky range flag of this sample: an activation beyond the f16 planes' range anywhere in its tiles (NaN compares false below too)
        if (A.ovf && !(gmax < kX2ActLimit)) atomicOr(A.ovf + b, 1);
    }
    ring.drain();
    H3D_TRACE(9);
    H3D_TRACE_DUMP(A.state);
}

size_t lds_bytes(const Args& A, int NT, int depth) {
    return sizeof(int) * (H3D_MAX_BLOCKS + 1) * kDescInts + sizeof(float) * ((size_t)((A.table_floats + 3) & ~3) + (size_t)A.n_ab * 2 * A.HdP + (size_t)A.n_cst * kShared + (A.heads ? 0 : 3 * (size_t)A.HdP)) +
           (size_t)depth * NT * 2048;
}

template <int NT, int DEPTH, bool SEG, bool X2, bool HEADS = false, bool MIDX3 = false>
int launch_seg(Args A, int B, int64_t groups, hipStream_t st) {
    H3D_ALLOW_MAX_LDS((synthesis_x3_kernel<NT, DEPTH, SEG, X2, HEADS, MIDX3>));
    A.n_tiles = (int)groups;
    if (A.tile_step <= 0) { A.tile_first = 0; A.tile_step = 1; }
    A.n_walk = A.tile_first < A.n_tiles ? (A.n_tiles - A.tile_first + A.tile_step - 1) / A.tile_step : 0;
    if (A.n_walk == 0) return H3D_OK;
    groups = A.n_walk;
    // persistent workgroups (one per CU at a time: registers and LDS): about four per CU in total, so that the tables are staged
    // once per ~n_tiles * B / (4 CUs) tiles while the tail of the launch stays short; segmented runs keep one tile per workgroup
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
    }
    static const int per_cu = getenv("H3D_SYNTH_WG_PER_CU") ? atoi(getenv("H3D_SYNTH_WG_PER_CU")) : 4;      // 0: one tile per workgroup
    // (a fallback launch -- run_if -- usually redoes ONE sample of the batch: that sample's workgroups must fill the chip by themselves)
    const int64_t per_sample = (SEG || per_cu <= 0) ? groups
                             : A.run_if ? std::min<int64_t>(groups, cus)
                             : std::max<int64_t>(1, std::min<int64_t>(groups, ((int64_t)per_cu * cus + B - 1) / B));
    h3d::pre_launch();
    hipLaunchKernelGGL((synthesis_x3_kernel<NT, DEPTH, SEG, X2, HEADS, MIDX3>), dim3((unsigned)per_sample, (unsigned)B), dim3(256),
                       lds_bytes(A, NT, DEPTH + (X2 ? 1 : 0)), st, A);
    return h3d::launch_status(X2 ? "h3d_synthesis_x2" : "h3d_synthesis_x3");
}

template <int NT, int DEPTH, bool X2>
int launch_one(const Args& A, int B, int64_t groups, hipStream_t st, bool heads = false) {
    // the state load/store paths are compiled only into the segmented variant (they cost registers)
    if (A.load_state || A.store_state) return launch_seg<NT, DEPTH, true, X2>(A, B, groups, st);
    if constexpr (X2) {
        if (A.mid_x3) return heads ? launch_seg<NT, DEPTH, false, true, true, true>(A, B, groups, st)
                                   : launch_seg<NT, DEPTH, false, true, false, true>(A, B, groups, st);
        if (heads) return launch_seg<NT, DEPTH, false, true, true>(A, B, groups, st);
    }
    return launch_seg<NT, DEPTH, false, X2>(A, B, groups, st);
}

}  // namespace

extern "C" int h3d_synthesis_x3_geometry_ok(int H, int W, int Hr, int Wr) {
    // 32 consecutive pixels of one image row must touch at most 8 low-res columns (x0 spread <= 6, conservative bound)
    return H >= 1 && Hr >= 1 && Wr >= 1 && W >= 32 && W % 32 == 0 && (int64_t)31 * Wr < (int64_t)6 * W;
}

static int synthesis_x(bool x2, const void* stream, int64_t total_stages, const float* tables, int table_floats,
                       const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                       const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                       float* state, int load_state, int store_state, h3d_stream_t stream_, int* ovf = nullptr,
                       const int* run_if = nullptr, int tile_first = 0, int tile_step = 1) {
    H3D_REQUIRE(stream && tables && desc && rgb, "h3d_synthesis_x3: null pointer");
    H3D_REQUIRE(h3d::aligned16(stream) && h3d::aligned16(tables), "h3d_synthesis_x3: stream/tables must be 16-byte aligned");
    H3D_REQUIRE(desc->n_blocks >= 1 && desc->n_blocks <= H3D_MAX_BLOCKS, "h3d_synthesis_x3: n_blocks=%d", desc->n_blocks);
    H3D_REQUIRE(B >= 0 && B <= 65535 && H >= 1 && W >= 1, "h3d_synthesis_x3: bad output shape");
    if (desc->C > 256) {
        h3d::set_error("h3d_synthesis_x3: width %d exceeds the 256 this engine keeps in registers (use h3d_synthesis)", desc->C);
        return H3D_EUNSUPPORTED;
    }
    const int NT = desc->C > 128 ? 8 : 4;
    int64_t want = 0;
    bool seen_skip = false, any_pixel = false, any_const = false;
    int first_skip = desc->n_blocks;
    for (int k = 0; k < desc->n_blocks; ++k) {
        if (desc->block[k].skip && !seen_skip) first_skip = k;
        if (seen_skip && !desc->block[k].skip) {
            h3d::set_error("h3d_synthesis_x3: a block without skip connection after the first skip block is not supported; "
                           "use h3d_synthesis");
            return H3D_EUNSUPPORTED;
        }
        seen_skip = seen_skip || desc->block[k].skip;
        for (int s = 0; s < 2; ++s) {
            const h3d_spade_desc& sp = desc->block[k].spade[s];
            if (sp.pixel_style) {
                any_pixel = true;
                if (seen_skip) {
                    h3d::set_error("h3d_synthesis_x3: a per-pixel-style block after the first skip block needs a third "
                                   "accumulator set; use h3d_synthesis");
                    return H3D_EUNSUPPORTED;
                }
                H3D_REQUIRE(sp.g_offset >= 0 && sp.g_offset + kShared <= g_channels && (sp.g_offset & 3) == 0,
                            "h3d_synthesis_x3: g_offset out of range");
                H3D_REQUIRE(sp.cst_index >= 0 && sp.cst_index < n_cst, "h3d_synthesis_x3: cst_index out of range");
                want += 16;
            } else {
                any_const = true;
                H3D_REQUIRE(sp.ab_index >= 0 && sp.ab_index < n_ab, "h3d_synthesis_x3: ab_index out of range");
            }
            want += 2 * NT;
        }
    }
    // per-pixel styles must form a leading run of whole blocks (both SPADEs), everything after it constant style
    int n_pixel_blocks = 0;
    while (n_pixel_blocks < desc->n_blocks && desc->block[n_pixel_blocks].spade[0].pixel_style &&
           desc->block[n_pixel_blocks].spade[1].pixel_style)
        ++n_pixel_blocks;
    for (int k = n_pixel_blocks; k < desc->n_blocks; ++k)
        if (desc->block[k].spade[0].pixel_style || desc->block[k].spade[1].pixel_style) {
            h3d::set_error("h3d_synthesis_x3: per-pixel-style SPADEs must be the leading whole blocks (block %d breaks that); "
                           "use h3d_synthesis_x3t / h3d_synthesis", k);
            return H3D_EUNSUPPORTED;
        }
    // x2 plans (round 6): a constant-style SPADE in front of the first skip block with g_offset == 1 has its convolution's stages in
    // the x3 format (bf16 hi | lo) and runs on three products; all of those blocks or none
    int mid_marked = 0, mid_total = 0;
    for (int k = n_pixel_blocks; k < first_skip; ++k)
        for (int s = 0; s < 2; ++s) { ++mid_total; mid_marked += desc->block[k].spade[s].g_offset == 1; }
    H3D_REQUIRE(mid_marked == 0 || (x2 && mid_marked == mid_total && !load_state && !store_state),
                "h3d_synthesis_x2: %d of %d middle-block convolutions are marked for the x3 format (all or none; single-launch x2 plans only)",
                mid_marked, mid_total);
    H3D_REQUIRE(want == total_stages, "h3d_synthesis_x3: stream has %lld stages, descriptor needs %lld",
                (long long)total_stages, (long long)want);
    H3D_REQUIRE(!any_pixel || (G && cst && (g_channels & 3) == 0 && h3d::aligned16(G)), "h3d_synthesis_x3: G/cst missing");
    if (any_pixel && !h3d_synthesis_x3_geometry_ok(H, W, Hr, Wr)) {
        h3d::set_error("h3d_synthesis_x3: %dx%d from %dx%d: the matrix-core resize needs W %% 32 == 0 and 31*Wr/W < 6; "
                       "use h3d_synthesis", H, W, Hr, Wr);
        return H3D_EUNSUPPORTED;
    }
    H3D_REQUIRE(!any_const || ab, "h3d_synthesis_x3: ab table missing");
    H3D_REQUIRE(store_state || desc->block[desc->n_blocks - 1].to_rgb || (x2 && desc->block[desc->n_blocks - 1].spade[1].b_conv >= 0),
                "h3d_synthesis_x3: the last block must feed ToRGB");
    for (int k = 0; k < desc->n_blocks; ++k) {
        const int64_t ho = desc->block[k].spade[1].b_conv;       // x2 plans: the ToRGB head table of a skip block (floats into `tables`)
        if (ho < 0) continue;
        H3D_REQUIRE(x2 && !load_state && !store_state && desc->block[k].skip && !desc->block[k].to_rgb && (ho & 3) == 0 &&
                        ho + (int64_t)2 * NT * 64 <= table_floats,
                    "h3d_synthesis_x2: bad ToRGB head table of block %d (offset %lld)", k, (long long)ho);
    }
    H3D_REQUIRE((!load_state && !store_state) || (state && h3d::aligned16(state)), "h3d_synthesis_x3: state buffer missing");
    if (B == 0) return H3D_OK;
    Args A{};
    A.state = state; A.load_state = load_state; A.store_state = store_state;
    A.ovf = ovf; A.run_if = run_if;
    A.mid_x3 = mid_marked > 0;
    H3D_REQUIRE(tile_first >= 0 && tile_step >= 1, "h3d_synthesis_x3: tile subset (%d, %d)", tile_first, tile_step);
    A.tile_first = tile_first; A.tile_step = tile_step;
    A.stream = static_cast<const unsigned char*>(stream);
    A.tables = tables; A.D = *desc; A.G = G; A.cst = cst; A.ab = ab; A.rgb = rgb;
    A.table_floats = table_floats; A.total_stages = (int)total_stages; A.g_channels = g_channels; A.Hr = Hr; A.Wr = Wr;
    if (!any_pixel) A.g_channels = 0;
    A.n_cst = n_cst; A.n_ab = n_ab; A.H = H; A.W = W; A.C = desc->C; A.HdP = NT * 32; A.first_skip = first_skip; A.n_pixel_blocks = n_pixel_blocks;
    for (int k = 0; k < desc->n_blocks; ++k) A.heads = A.heads || desc->block[k].spade[1].b_conv >= 0;      // validated above: x2, single launch
    const int extra = x2 ? 1 : 0;                                      // x2 keeps one more ring buffer (WeightRing LAG = 1)
    const bool deep = lds_bytes(A, NT, 6 + extra) <= 160 * 1024;      // deepest weight ring the tables leave room for
    if (lds_bytes(A, NT, kRingDepth + extra) > 160 * 1024) {
        h3d::set_error("h3d_synthesis_x3 / _x2: tables (%d floats) + ring do not fit the 160 KB LDS; use h3d_synthesis", table_floats);
        return H3D_EUNSUPPORTED;
    }
    const int64_t groups = ((int64_t)H * W + 127) / 128;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_synthesis_x3: image too large");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const bool heads = A.heads != 0;                                 // x2 plans: ToRGB head tables present
#ifdef H3D_DEV_ONLY_HOT       // development: compile only the instantiation the BASELINE cfg-3 bench runs (fast ISA / resource turnaround)
    (void)deep;
    return A.mid_x3 ? launch_seg<8, kRingDepth, false, true, true, true>(A, B, groups, st)
         : heads ? launch_seg<8, kRingDepth, false, true, true>(A, B, groups, st) : launch_seg<8, kRingDepth, false, true>(A, B, groups, st);
#else
    if (x2) {
        if (NT == 8) return deep ? launch_one<8, 6, true>(A, B, groups, st, heads) : launch_one<8, kRingDepth, true>(A, B, groups, st, heads);
        return deep ? launch_one<4, 6, true>(A, B, groups, st, heads) : launch_one<4, kRingDepth, true>(A, B, groups, st, heads);
    }
    if (NT == 8) return deep ? launch_one<8, 6, false>(A, B, groups, st) : launch_one<8, kRingDepth, false>(A, B, groups, st);
    return deep ? launch_one<4, 6, false>(A, B, groups, st) : launch_one<4, kRingDepth, false>(A, B, groups, st);
#endif
}

extern "C" int h3d_synthesis_x3(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                                const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                                const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                                float* state, int load_state, int store_state, h3d_stream_t stream_) {
    return synthesis_x(false, stream, total_stages, tables, table_floats, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W,
                       state, load_state, store_state, stream_);
}
extern "C" int h3d_synthesis_x2(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                                const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                                const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                                float* state, int load_state, int store_state, h3d_stream_t stream_) {
    return synthesis_x(true, stream, total_stages, tables, table_floats, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W,
                       state, load_state, store_state, stream_);
}
/* Range-guarded pair (round 4).  h3d_synthesis_x2_guarded is h3d_synthesis_x2 that also ORs 1 into *overflow (device memory,
 * zeroed by the caller) when any activation it converted to the f16 planes was >= 2^15 in magnitude or non-finite -- its image
 * is then not to be used.  h3d_synthesis_x3_if is h3d_synthesis_x3 (bf16 planes: fp32 exponent range) that returns at once,
 * leaving rgb untouched, when *run_if == 0.  Launched back to back on one stream with the same flag they give "x2, redone on
 * x3 when out of range" without a host synchronisation. */
extern "C" int h3d_synthesis_x2_guarded(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                                        const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                                        const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                                        int* overflow, h3d_stream_t stream_) {
    H3D_REQUIRE(overflow, "h3d_synthesis_x2_guarded: null flag");
    return synthesis_x(true, stream, total_stages, tables, table_floats, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W,
                       nullptr, 0, 0, stream_, overflow, nullptr);
}
extern "C" int h3d_synthesis_x3_if(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                                   const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                                   const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                                   const int* run_if, h3d_stream_t stream_) {
    H3D_REQUIRE(run_if, "h3d_synthesis_x3_if: null flag");
    return synthesis_x(false, stream, total_stages, tables, table_floats, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W,
                       nullptr, 0, 0, stream_, nullptr, run_if);
}
/* Sampled error monitor of the x2 engine (round 5).  h3d_synthesis_x3_tiles is h3d_synthesis_x3 restricted to the 128-pixel
 * tiles tile_first, tile_first + tile_step, .. of every sample: it writes those pixels of `rgb` (a scratch image of the full
 * shape) and nothing else.  h3d_synthesis_check compares an image with such a scratch image on exactly those tiles. */
extern "C" int h3d_synthesis_x3_tiles(const void* stream, int64_t total_stages, const float* tables, int table_floats,
                                      const h3d_synth_desc* desc, const float* G, int g_channels, int Hr, int Wr,
                                      const float* cst, int n_cst, const float* ab, int n_ab, float* rgb, int B, int H, int W,
                                      int tile_first, int tile_step, h3d_stream_t stream_) {
    return synthesis_x(false, stream, total_stages, tables, table_floats, desc, G, g_channels, Hr, Wr, cst, n_cst, ab, n_ab, rgb, B, H, W,
                       nullptr, 0, 0, stream_, nullptr, nullptr, tile_first, tile_step);
}

namespace {
// Stage 1, grid (B, kCheckSlices): slice s of sample b takes every kCheckSlices-th float4 of the three channel planes (max |img|:
// the scale the parity budget is relative to; img and ref agree to ~1e-3, so either serves) and every kCheckSlices-th sampled tile
// (max |img - ref|); the six maxima are merged with atomicMax on the bit patterns (non-negative floats order like unsigned ints, a
// NaN pattern sorts above infinity and so survives).  work [B][6] is zeroed by the entry point.
constexpr int kCheckSlices = 32;
__global__ __launch_bounds__(256) void synthesis_check_partial(const float* __restrict__ img, const float* __restrict__ ref, int64_t HW,
                                                               int n_tiles, int tile_first, int tile_step, unsigned* work) {
    __shared__ float red[6][4];
    const int b = blockIdx.x, sl = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fold = [](float a, float v) { return (a != a) ? a : (v != v) ? v : fmaxf(a, v); };   // keep a NaN once seen (fmaxf drops it)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = img + ((int64_t)b * 3 + c) * HW;
        const int64_t n4 = (HW % 4 == 0 && (reinterpret_cast<size_t>(p) & 15) == 0) ? HW / 4 : 0;
        for (int64_t i = (int64_t)sl * 256 + t; i < n4; i += (int64_t)kCheckSlices * 256) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            m[c] = fold(fold(fold(fold(m[c], fabsf(v.x)), fabsf(v.y)), fabsf(v.z)), fabsf(v.w));
        }
        for (int64_t i = n4 * 4 + (int64_t)sl * 256 + t; i < HW; i += (int64_t)kCheckSlices * 256) m[c] = fold(m[c], fabsf(p[i]));
    }
    for (int64_t tile = tile_first + (int64_t)sl * tile_step; tile < n_tiles; tile += (int64_t)kCheckSlices * tile_step) {
        for (int i = t; i < 128; i += 256) {
            const int64_t p = (int64_t)tile * 128 + i;
            if (p >= HW) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                m[3 + c] = fold(m[3 + c], fabsf(img[((int64_t)b * 3 + c) * HW + p] - ref[((int64_t)b * 3 + c) * HW + p]));
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[k] = fold(m[k], __shfl_xor(m[k], o, 64));
        if (lane == 0) red[k][wave] = m[k];
    }
    __syncthreads();
    if (t < 6) {
        const float v = fold(fold(red[t][0], red[t][1]), fold(red[t][2], red[t][3]));
        atomicMax(work + b * 6 + t, __float_as_uint(v));
    }
}
// Stage 2, one thread per sample: err = max_c (diff_c / scale_c); anything not finite -> infinity; flag when err > tol
__global__ void synthesis_check_final(const unsigned* __restrict__ work, int B, float tol, int* flag, float* err_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float err = 0.f;
    bool bad = false;
    for (int c = 0; c < 3; ++c) {
        const float r = __uint_as_float(work[b * 6 + c]), d = __uint_as_float(work[b * 6 + 3 + c]);
        bad = bad || !(r <= 3.0e38f) || !(d <= 3.0e38f);
        err = fmaxf(err, d / fmaxf(r, 1e-30f));
    }
    if (bad) err = __builtin_inff();
    if (err_out) err_out[b] = err;
    if (!(err <= tol)) atomicOr(flag + b, 1);
}
}  // namespace

extern "C" int h3d_synthesis_check(const float* rgb, const float* rgb_ref, int B, int H, int W, int tile_first, int tile_step,
                                   float tol, int* flag, float* err_out, float* work, h3d_stream_t stream_) {
    H3D_REQUIRE(rgb && rgb_ref && flag && work, "h3d_synthesis_check: null pointer");
    H3D_REQUIRE(B >= 0 && B <= 65535 && H >= 1 && W >= 1 && tile_first >= 0 && tile_step >= 1 && tol >= 0.f, "h3d_synthesis_check: bad arguments");
    if (B == 0) return H3D_OK;
    const int64_t HW = (int64_t)H * W;
    const int n_tiles = (int)((HW + 127) / 128);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (hipMemsetAsync(work, 0, sizeof(float) * 6 * (size_t)B, st) != hipSuccess) {
        h3d::set_error("h3d_synthesis_check: hipMemsetAsync failed");
        return H3D_ELAUNCH;
    }
    h3d::pre_launch();
    hipLaunchKernelGGL(synthesis_check_partial, dim3((unsigned)B, kCheckSlices), dim3(256), 0, st, rgb, rgb_ref, HW, n_tiles, tile_first,
                       tile_step, reinterpret_cast<unsigned*>(work));
    int rc = h3d::launch_status("h3d_synthesis_check");
    if (rc) return rc;
    hipLaunchKernelGGL(synthesis_check_final, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, reinterpret_cast<const unsigned*>(work), B, tol,
                       flag, err_out);
    return h3d::launch_status("h3d_synthesis_check");
}

/* HOST helper: LDS bytes of the x3 (x2 = 0) / x2 (x2 = 1) kernel at its minimum ring depth for a network of width C with
 * `table_floats` static table floats, n_ab constant-style and n_cst per-pixel SPADEs -- the planner's fit test (<= 160 KiB). */
extern "C" int64_t h3d_synthesis_x3_lds_bytes(int table_floats, int n_ab, int n_cst, int C, int x2) {
    Args A{};
    A.table_floats = table_floats; A.n_ab = n_ab; A.n_cst = n_cst;
    A.heads = (x2 & 2) != 0;                 // x2 = 1: the x2 engine; x2 = 3: an x2 plan with ToRGB head tables (no zero table in LDS)
    x2 &= 1;
    const int NT = C > 128 ? 8 : 4;
    A.HdP = NT * 32;
    return (int64_t)lds_bytes(A, NT, kRingDepth + (x2 ? 1 : 0));
}
/* LDS the x2 variant needs beyond the x3 one, for host-side planning (one more ring buffer) */
extern "C" int h3d_synthesis_x2_extra_lds(int C) { return (C > 128 ? 8 : 4) * 2048; }
