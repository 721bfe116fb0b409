// Split-operand ("x3") matrix-core engine with the activations of a 64-sample tile resident in LDS ("x3t"), gfx950.
//
// Why a second x3 engine: x3_common.hpp keeps a sample's activations in the registers of ONE lane (a wave owns 32
// samples and all channels), which needs 2 x C accumulator registers per lane and stops at C = 256.  Here the N
// (output feature) dimension is split over the four waves of a workgroup instead, so the width is bounded by LDS, not
// by registers (C <= 448 with 64 samples per workgroup): MAP3DBN (384) and MAP3DBN512L (420, the released checkpoint's
// architecture) run on the f16/bf16 matrix cores instead of the 16x slower fp32 ones.
//
//   arithmetic   x = hi + lo (f16 for the field, bf16 for the synthesis network), product = hi*hi + hi*lo + lo*hi on
//                v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulation -- identical to x3_common.hpp.
//   dataflow     D[feature][sample] = W[feature][k] * X[k][sample]: weights are the A operand and come straight from
//                L2 in A-fragment order (each wave streams only the tiles it owns, four k-steps ahead, in registers);
//                activations are the B operand and live in LDS as ready-made hi / lo B fragments
//                    actT[mt][ks][plane][64 lanes][8 x 16 bit]          (2 KB per sample tile mt and k-step ks)
//                shared by the four waves (one ds_read_b128 per fragment, conflict-free).
//   K order      the accumulator registers a lane holds for feature tile t are, in order, the B-fragment elements of
//                k-steps 2t and 2t+1 ("accumulator order", the same permutation as x3_common.hpp):
//                    feature(ks, h, e) = 32*(ks/2) + (e & 3) + 8*(2*(ks & 1) + (e >> 2)) + 4*h
//                so the epilogue of a unit writes its fragments with two lane-local ds_write_b128 per plane, no
//                transposition; the host packs the K dimension of every matrix fed that way accordingly.  Matrices
//                fed from memory (coordinates, geometry features, view direction, shared-MLP activations) use the
//                natural order k = 16*ks + 8*h + e.
//   work split   a *unit* is (feature tile nt, sample tile mt).  Wave w owns the full tiles nt = w, w+4, .. (< 4*NTF)
//                for both sample tiles and, when the tile count is 4*NTF + 2, one extra unit
//                (nt = 4*NTF + (w >> 1), mt = w & 1): 2*NTF + NX accumulator tiles per wave, perfectly balanced for the
//                even tile counts the hosts pad to (12 for width 384, 14 for 420).
#pragma once
#include "x3_common.hpp"
#include <string.h>

namespace h3d {

constexpr int kX3tMT = 2;            // sample tiles (of 32) per workgroup
#ifndef H3D_X3T_DEPTH
#define H3D_X3T_DEPTH 4
#endif
constexpr int kX3tDepth = H3D_X3T_DEPTH;        // weight k-steps in flight per wave (register ring)

// byte offset of fragment plane (mt, ks, plane) inside an activation tile with KS k-steps per sample tile
__device__ __forceinline__ int x3t_frag(int KS, int mt, int ks, int plane) { return ((mt * KS + ks) * 2 + plane) * 1024; }

__host__ __device__ inline int x3t_acc_k(int ks, int hh, int e) { return 32 * (ks / 2) + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hh; }

// Which units a wave owns.
template <int NTF, int NX>
struct X3tUnits {
    static constexpr int NU = 2 * NTF + NX;       // accumulator tiles per wave
    static constexpr int NA = NTF + NX;           // weight tiles per wave and k-step
    int nt[NA];                                   // feature tile of weight slot i (slot NTF = the extra unit's)
    int xmt;                                      // sample tile of the extra unit
    __device__ __forceinline__ void init(int wave) {
#pragma unroll
        for (int i = 0; i < NTF; ++i) nt[i] = wave + 4 * i;
        if constexpr (NX > 0) nt[NA - 1] = 4 * NTF + (wave >> 1);
        xmt = wave & 1;
    }
    // unit u -> (weight slot, sample tile): u = 2*i + mt for the full tiles, u = 2*NTF for the extra unit
    static __device__ __forceinline__ constexpr int slot(int u) { return u < 2 * NTF ? u / 2 : NTF; }
    __device__ __forceinline__ int tile(int u) const { return nt[slot(u)]; }
    __device__ __forceinline__ int mt(int u) const { return u < 2 * NTF ? (u & 1) : xmt; }
};

// One 16-byte weight-fragment load, issued through inline asm on purpose.  With compiler-visible loads hipcc's waitcnt
// pass loses count at the loop back-edge of the register ring below and, once per trip, waits for everything but the
// newest k-step (s_waitcnt vmcnt(6) where vmcnt(18) is right): the prefetch distance collapses from three k-steps to
// one.  The ring is synchronised by hand instead (x3t_wait_frags): loads retire in order, so "at most N outstanding"
// is exact by construction, and loads the compiler issues on its own only make the wait stricter, never weaker.
// Address = uniform base (SGPR pair: the tile's k-step) + per-lane byte offset (one VGPR shared by every load) + OFF.
template <int OFF>
__device__ __forceinline__ void x3t_gload(u32x4& dst, const unsigned char* base, unsigned lane_off) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
// s_waitcnt vmcnt(N), tied to the fragments about to be consumed: their users read the post-wait values, so the
// scheduler cannot move an MFMA above the wait.
template <int N, int NA, bool LO = true>
__device__ __forceinline__ void x3t_wait_frags(u32x4 (&h)[NA], u32x4 (&l)[NA]) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(h[0]) : "n"(N) : "memory");
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if (i) asm volatile("" : "+v"(h[i]));
        if (LO) asm volatile("" : "+v"(l[i]));
    }
}

// The weight-fragment ring of one wave: kX3tDepth k-steps of (hi, lo) A fragments for its NA tiles.  Owned by the caller
// so that the first D-1 k-steps of the NEXT GEMM can be requested (x3t_prefetch) before the epilogue / barriers that
// separate two GEMMs: their L2 latency then hides behind that work instead of stalling the GEMM's first MFMA.
template <int NA>
struct X3tRing {
    struct AF { u32x4 h[NA], l[NA]; };
    AF a[kX3tDepth];
};

// Weight layouts of a matrix, per tile (x3t_pack_f16 / x3t_pack_x2):
//   x3 format   [k-step][hi | lo][64 lanes][16 B]: 2 KiB per k-step (tiers P = 1, 2, 3; the short input phases of every tier)
//   x2c format  (round 6, tier P = 4) per K-tile T (k-steps 2T, 2T + 1) 3 KiB:
//                   +0     f16 hi fragment of k-step 2T
//                   +1024  lo record, 16 B per lane: the 16 six-bit e2m3 codes of lo * 2^12 * alpha (dwords 3-5 of the fp6 A
//                          operand) and the lane's block-scale dword (the e8m0 byte of 1 / alpha in bits 0-7, zeros above)
//                   +2048  f16 hi fragment of k-step 2T + 1
//               The 16 hi codes of the fp6 operand (dwords 0-2) are NOT stored: the kernel converts them from the two f16 hi
//               fragments it holds anyway (v_cvt_scalef32_pk32_fp6_f16 with the lane's scale) -- 3 bytes per weight through the
//               vector-memory path instead of 4.  Measured (profiles/r6_wide_*): the engine's GEMM loops run at the 64 B/clk/CU of
//               that path (28 KiB of weights per k-step and 64-sample tile at width 448 against 336 matrix-pipe cycles).
//               A trailing odd k-step of a matrix (the colour layer's view-direction k-step) keeps the x3 format.
template <bool X2C>
__host__ __device__ inline int64_t x3t_tile_bytes(int KStot) { return X2C ? (int64_t)(KStot >> 1) * 3072 + (KStot & 1) * 2048 : (int64_t)KStot * 2048; }
template <bool X2C>
__host__ __device__ inline int x3t_kstep_off(int ks) { return X2C ? (ks >> 1) * 3072 + (ks & 1) * 2048 : ks * 2048; }

// x2c, odd k-step (gemm_x3t): memory operation `op` is issued behind matrix instruction j(op), the j with
// j * NOPS / NM <= op < (j + 1) * NOPS / NM.  The fp6 instructions are j = 0 .. NU - 1; those of weight tile i are j = 2 i, 2 i + 1
// (the extra unit's: 2 NTF).  Operation NBH + i refills weight tile i's hi fragment: not before the tile's last fp6 instruction;
// operations >= NBH + NA read / load record halves: not before the last fp6 instruction of the k-step.
template <int NTF, int NX>
constexpr bool x2c_order_ok(int NBH, int NOPS, int NM) {
    auto behind = [&](int op) { int j = 0; while ((j + 1) * NOPS / NM <= op) ++j; return j; };
    const int NA = NTF + NX, NU = 2 * NTF + NX;
    for (int i = 0; i < NA; ++i)
        if (behind(NBH + i) < (i < NTF ? 2 * i + 1 : 2 * NTF)) return false;
    return behind(NBH + NA) >= NU - 1;
}

// Request the first D-1 k-steps of a phase (starting at byte `phase_off` of every tile, tiles `tile_stride` bytes apart) of the
// wave's tiles of matrix W into ring slots 0 .. D-2.  Between this call and the GEMM that consumes it (PRE = true) the caller must
// not start another GEMM on the same ring.  P = 4: the phase is in the x2c format and starts on a K-tile boundary.
template <int NTF, int NX, int P = 3>
__device__ __forceinline__ void x3t_prefetch(X3tRing<NTF + NX>& R, const unsigned char* __restrict__ W, int64_t tile_stride, int phase_off,
                                             const X3tUnits<NTF, NX>& U, int lane) {
    constexpr int NA = NTF + NX;
    constexpr bool X2 = P == 4;
    const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll
    for (int d = 0; d < kX3tDepth - 1; ++d)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned char* q = W + (int64_t)U.nt[i] * tile_stride + phase_off + x3t_kstep_off<X2>(d);
            x3t_gload<0>(R.a[d].h[i], q, lane_off);
            if constexpr (X2) { if ((d & 1) == 0) x3t_gload<1024>(R.a[d].l[i], q, lane_off); }      // the K-tile's lo record rides with its even k-step
            else if constexpr (P >= 2) x3t_gload<1024>(R.a[d].l[i], q, lane_off);
        }
}

// fp6 A operand of a K-tile (x2c format): dwords 0-2 = the hi codes, converted here from the two f16 hi fragments with the lane's
// block scale; dwords 3-5 = the lo codes and dword 6 = the scale, as loaded (lr).  (The upper 16 inputs of the conversion are
// don't-cares: its dwords 3-5 are discarded.)
__device__ __forceinline__ i32x8 x2c_record(const u32x4& h0, const u32x4& h1, const u32x4& lr) {
    typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
    const F16::vec8 a = __builtin_bit_cast(F16::vec8, h0), b = __builtin_bit_cast(F16::vec8, h1);
    const f16x16 hi = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const f16x32 v = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                             -1, -1, -1, -1, -1, -1);            // upper half undefined: no copies
    const float cs = __builtin_bit_cast(float, lr[3] << 23);                    // 1 / alpha = 2^(byte - 127): the dword holds the byte alone
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, cs);
    return i32x8{(int)r[0], (int)r[1], (int)r[2], (int)lr[0], (int)lr[1], (int)lr[2], (int)lr[3], 0};
}

// acc[u] += W(tile(u), ks) x X(mt(u), ks), ks = 0 .. KS-1 (KS a multiple of kX3tDepth unless GUARD).
//   bT     LDS: fragment plane (mt 0, first k-step of this phase, hi); sample tiles are mt_stride bytes apart
//   W      global: A fragments [tile][KStot][plane][64][16 B]; this phase uses k-steps ks0 .. ks0+KS-1 of every tile
//   SWAP   D[sample][feature] instead (activations as the A operand): used by the feature head of the fused render,
//          where the weighted sum over the samples of a ray becomes a sum over accumulator registers.
// Register ring: the A fragments of k-step s+3 are requested before the MFMAs of k-step s (four k-steps in flight,
// ~2000 matrix-pipe cycles: an L2 miss served by the Infinity Cache is covered), B fragments one k-step ahead.
// GUARD: a short phase (input layers with KS = 1 or 2 k-steps); otherwise KS is a multiple of the ring depth, >= depth.
// PRE: the ring already holds the requests of x3t_prefetch(R, W, KStot, ks0, ..) (full phases only).
// P: partial products per operand pair -- the precision tier (same data layout, fewer planes touched):
//      3  hi*hi + hi*lo + lo*hi   fp32-class (22 / 16 significant bits per operand with f16 / bf16 halves)
//      2  hi*hi + lo*hi           weights to full precision, activations rounded to ONE 16-bit value
//      1  hi*hi                   plain f16 / bf16 matrix-core arithmetic
//      4  "x2" (x3_common.hpp)    hi*hi on the f16 matrix cores + ONE block-scaled fp6 instruction per K-tile (two k-steps) for
//                                 both cross terms.  The "lo" planes of weights and activations then hold the halves of the
//                                 K-tile's fp6 records (dwords 0-3 with the even k-step, 4-7 with the odd one): an even k-step
//                                 issues NU f16 instructions, an odd one NU fp6 instructions (which read the record halves of
//                                 this AND the previous k-step) followed by NU f16 instructions.  The previous k-step's ring
//                                 slots are the ones this k-step refills, so every load / read into a record half is pinned
//                                 behind the last fp6 instruction (operation order: weight hi, fragment hi, fragment lo, weight lo).
template <typename T, int NTF, int NX, bool SWAP, bool GUARD = false, bool PRE = false, int P = 3>
__device__ __forceinline__ void gemm_x3t(f32x16 (&acc)[2 * NTF + NX], const unsigned char* bT, int mt_stride,
                                         const unsigned char* __restrict__ W, int64_t tile_stride, int phase_off, int KS,
                                         const X3tUnits<NTF, NX>& U, int lane, X3tRing<NTF + NX>& R) {
    constexpr int NA = NTF + NX, NU = 2 * NTF + NX, D = kX3tDepth;
    constexpr bool X2 = P == 4;
    constexpr bool WLO = P >= 2, XLO = P >= 3;       // which lo planes are read
    constexpr int NLA = (WLO ? 2 : 1) * NA;          // weight loads per k-step
    static_assert(P >= 1 && P <= 4, "1, 2 or 3 partial products, or 4 = x2");
    static_assert(!X2 || (D == 4 && !GUARD), "x2: k-step parity follows the ring slot (the in-flight counts below assume depth 4); short phases stay on three products");
    // loads that may stay in flight while waiting for a k-step's fragments: those of the D - 2 k-steps behind it -- x2c: one even
    // k-step (hi + lo record: 2 NA loads) and one odd one (hi: NA loads) per pair
    constexpr int kInflight = X2 ? (D - 2) / 2 * 3 * NA : (D - 2) * NLA;
    static_assert(!(GUARD && PRE), "short phases load everything themselves");
    typedef typename X3tRing<NA>::AF AF;
    struct BF { u32x4 h[2], l[2], xh, xl; };
    AF (&a)[D] = R.a;
    const unsigned char* wp[NA];              // uniform: first k-step of this phase of the wave's tiles
#pragma unroll
    for (int i = 0; i < NA; ++i) wp[i] = W + (int64_t)U.nt[i] * tile_stride + phase_off;
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned char* bp = bT + lane * 16;
    const unsigned char* bx = bp + U.xmt * mt_stride;
    auto loadA = [&](AF& f, int ks) __attribute__((always_inline)) {      // all loads of k-step ks (x2c: the lo record rides with the even one)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned char* q = wp[i] + x3t_kstep_off<X2>(ks);     // one scalar address per tile and k-step; lo plane / record 1 KB further
            x3t_gload<0>(f.h[i], q, lane_off);
            if constexpr (X2) { if ((ks & 1) == 0) x3t_gload<1024>(f.l[i], q, lane_off); }
            else if constexpr (WLO) x3t_gload<1024>(f.l[i], q, lane_off);
        }
    };
    auto loadB = [&](BF& f, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (NTF > 0) {
                f.h[mt] = *reinterpret_cast<const u32x4*>(bp + mt * mt_stride + ks * 2048);
                if constexpr (XLO) f.l[mt] = *reinterpret_cast<const u32x4*>(bp + mt * mt_stride + ks * 2048 + 1024);
            }
        }
        if (NX) {
            f.xh = *reinterpret_cast<const u32x4*>(bx + ks * 2048);
            if constexpr (XLO) f.xl = *reinterpret_cast<const u32x4*>(bx + ks * 2048 + 1024);
        }
    };
    // MFMA j of a k-step, j = 0 .. P*NU-1: pass j / NU (hi*hi, then lo*hi, then hi*lo) over all units -- consecutive MFMAs
    // never share an accumulator.  x2 (ODD k-step): j < NU are the fp6 instructions of the K-tile (record halves of the
    // previous k-step ap / bp and of this one), j >= NU the f16 ones; even k-step: f16 only.
    i32x8 wrec[NA];                           // x2: the fp6 A operands of the current K-tile (live from a tile's first fp6 instruction to its last)
    auto mfma1 = [&](auto jc, auto oddc, const AF& a, const BF& b, const AF& ap, const BF& bp) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr bool ODD = decltype(oddc)::value != 0;
        if constexpr (X2) {
            if constexpr (ODD && j < NU) {
                constexpr int u = j, sl = u < 2 * NTF ? u / 2 : NTF;
                // the K-tile's weight record: hi codes converted from the two hi fragments (previous, even k-step: ap.h; this one:
                // a.h), lo codes + scale from the record loaded with the even k-step (ap.l) -- built once per tile, in front of its
                // first instruction
                if constexpr (u >= 2 * NTF || (u & 1) == 0) wrec[sl] = x2c_record(ap.h[sl], a.h[sl], ap.l[sl]);
                const u32x4 x0 = u < 2 * NTF ? bp.l[u & 1] : bp.xl, x1 = u < 2 * NTF ? b.l[u & 1] : b.xl;
                const i32x8 x6 = {(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                acc[u] = mm6<SWAP>(wrec[sl], x6, acc[u]);
            } else {
                constexpr int u = ODD ? j - NU : j, sl = u < 2 * NTF ? u / 2 : NTF;
                const u32x4 xv = u < 2 * NTF ? b.h[u & 1] : b.xh;
                acc[u] = mm<T, SWAP>(__builtin_bit_cast(typename T::vec8, a.h[sl]), __builtin_bit_cast(typename T::vec8, xv), acc[u]);
            }
        } else {
            constexpr int p = j / NU, u = j % NU, sl = u < 2 * NTF ? u / 2 : NTF;
            const u32x4 wv = p == 1 ? a.l[sl] : a.h[sl];
            const u32x4 xv = u < 2 * NTF ? (p == 2 ? b.l[u & 1] : b.h[u & 1]) : (p == 2 ? b.xl : b.xh);
            acc[u] = mm<T, SWAP>(__builtin_bit_cast(typename T::vec8, wv), __builtin_bit_cast(typename T::vec8, xv), acc[u]);
        }
    };
    auto mfmas = [&](const AF& a, const BF& b) __attribute__((always_inline)) {      // short (GUARD) phases: never x2
        // (a non-generic lambda's body is instantiated with the enclosing template even where `if constexpr (GUARD)` never calls
        // it: without this guard the x2 instantiations indexed acc[] with j up to P * NU - 1 -- dead code, but out-of-bounds code)
        if constexpr (!X2) static_for<0, P * NU>([&](auto jc) __attribute__((always_inline)) { mfma1(jc, IC<0>{}, a, b, a, b); });
    };
    // One k-step with its memory traffic in the shadow of the matrix pipe (one wave per SIMD: whatever is issued between
    // two MFMAs is free, whatever is issued before the first one leaves the pipe idle): after MFMA j comes weight load j
    // (tile j/2, plane j%2) of k-step `ka`, then the LDS reads of k-step `kb`, one per MFMA.  Every position is pinned.
    // (LA / LB: whether this k-step still requests weights / fragments; references, never pointers: taking the address of a
    // ring element would move the whole ring to scratch memory)
    // x2: a_nxt / b_nxt are the slots of the PREVIOUS k-step; an odd k-step reads their record halves in its fp6 instructions
    // before anything is loaded into them (operation order below).
    auto kstep = [&](auto la, auto lb, auto oddc, const AF& a_cur, const BF& b_cur, AF& a_nxt, int ka, BF& b_nxt, int kb) __attribute__((always_inline)) {
        constexpr bool LA = decltype(la)::value != 0, LB = decltype(lb)::value != 0, ODD = decltype(oddc)::value != 0;
        constexpr int XP = XLO ? 2 : 1;                            // fragment planes read per sample tile
        // x2c: this k-step requests k-step `ka` of the OTHER parity (the ring depth is even): an odd k-step requests an even one
        // (hi fragments + lo records: 2 NA loads), an even k-step an odd one (hi fragments: NA loads)
        constexpr int NLK = X2 ? (ODD ? 2 * NA : NA) : NLA;
        constexpr int NBF = NTF > 0 ? 2 * XP : 0, NB = NBF + (NX ? XP : 0), NOPS = NLK + NB;
        constexpr int NM = X2 ? (ODD ? 2 * NU : NU) : P * NU;
        constexpr int NBH = (NTF > 0 ? 2 : 0) + (NX ? 1 : 0);      // x2: fragment hi reads (= lo reads)
        auto wload = [&](auto tc, auto pc) __attribute__((always_inline)) {
            constexpr int tile = decltype(tc)::value, plane = decltype(pc)::value;
            if constexpr (LA) {
                // (x2c: ka has the parity opposite to this k-step's, so its offset inside the K-tile is a compile-time constant)
                const unsigned char* q = X2 ? wp[tile] + (ka >> 1) * 3072 + (ODD ? 0 : 2048) : wp[tile] + ka * 2048;
                if constexpr (plane == 0) x3t_gload<0>(a_nxt.h[tile], q, lane_off);
                else x3t_gload<1024>(a_nxt.l[tile], q, lane_off);
            }
        };
        auto bread = [&](auto rc, auto pc) __attribute__((always_inline)) {     // r: 0, 1 = sample tiles of the full units, 2 = extra unit
            constexpr int r = decltype(rc)::value, plane = decltype(pc)::value;
            if constexpr (LB) {
                if constexpr (r < 2) {
                    const unsigned char* q = bp + r * mt_stride + kb * 2048 + plane * 1024;
                    if constexpr (plane == 0) b_nxt.h[r] = *reinterpret_cast<const u32x4*>(q);
                    else b_nxt.l[r] = *reinterpret_cast<const u32x4*>(q);
                } else {
                    if constexpr (plane == 0) b_nxt.xh = *reinterpret_cast<const u32x4*>(bx + kb * 2048);
                    else b_nxt.xl = *reinterpret_cast<const u32x4*>(bx + kb * 2048 + 1024);
                }
            }
        };
        // memory operation i = 0 .. NOPS-1
        auto memop = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (X2) {
                // fragment hi (NBH), weight hi (NA), fragment lo = record halves (NBH), weight lo records (NA, odd k-steps only).
                // In an odd k-step the slots being refilled (a_nxt, b_nxt) are the previous, even k-step's, whose weight hi
                // fragments, weight lo records and fragment record halves feed this k-step's fp6 instructions: the refill of
                // weight tile i comes behind the tile's last fp6 instruction, every read / load into a record (half) behind the
                // LAST fp6 instruction (x2c_order_ok, checked where the operations are placed)
                if constexpr (i < NBH) bread(IC<(NTF > 0 ? i : 2)>{}, IC<0>{});
                else if constexpr (i < NBH + NA) wload(IC<i - NBH>{}, IC<0>{});
                else if constexpr (i < 2 * NBH + NA) bread(IC<(NTF > 0 ? i - NBH - NA : 2)>{}, IC<1>{});
                else wload(IC<i - 2 * NBH - NA>{}, IC<1>{});
            } else if constexpr (i < NLA) {
                constexpr int tile = WLO ? i / 2 : i, plane = WLO ? i % 2 : 0;
                wload(IC<tile>{}, IC<plane>{});
            } else {
                constexpr int r = i - NLA;
                if constexpr (r < NBF) bread(IC<r / XP>{}, IC<r % XP>{});
                else bread(IC<2>{}, IC<r - NBF>{});
            }
        };
        if constexpr (X2 && ODD) {
            static_assert(x2c_order_ok<NTF, NX>(NBH, NOPS, NM), "x2c: a refill would overtake the fp6 instructions that read its slot");
        }
        static_for<0, NM>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            mfma1(jc, oddc, a_cur, b_cur, a_nxt, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
            // operations [j*NOPS/NM, (j+1)*NOPS/NM) ride behind MFMA j: spread evenly over the k-step (with one product
            // per operand pair there are more memory operations than MFMAs: some gaps carry two)
            static_for<j * NOPS / NM, (j + 1) * NOPS / NM>([&](auto ic) __attribute__((always_inline)) {
                memop(ic);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    BF b[2];
    // (every load the caller issued before this point is older than the ring's loads: it cannot weaken the waits)
    if constexpr (GUARD) {
        // short phase (KS = 1 or 2 <= D-1): everything up front, one wait
        static_assert(D >= 3, "short phases keep all their k-steps in the ring");
        loadA(a[0], 0);
        loadB(b[0], 0);
        if (KS > 1) { loadA(a[1], 1); loadB(b[1], 1); }
        x3t_wait_frags<0, NA, WLO>(a[0].h, a[0].l);
        mfmas(a[0], b[0]);
        if (KS > 1) {
            x3t_wait_frags<0, NA, WLO>(a[1].h, a[1].l);
            mfmas(a[1], b[1]);
        }
    } else {
        if constexpr (!PRE) {
#pragma unroll
            for (int d = 0; d < D - 1; ++d) loadA(a[d], d);
        }
        loadB(b[0], 0);
        // slot d of a trip: the fragments of its k-step were requested D-1 slots ago; the requests of the D-2 slots in
        // between -- (D-2) * NLA loads -- may stay in flight.  Its own requests (k-step +D-1) follow inside kstep.
        for (int ks = 0; ks < KS - D; ks += D) {
            static_for<0, D>([&](auto dc) __attribute__((always_inline)) {
                constexpr int d = decltype(dc)::value;
#ifdef H3D_EXPERIMENT_TRACE_FINE
                H3D_TRACE(50 + d);                    // before the wait for k-step ks + d
#endif
                x3t_wait_frags<kInflight, NA, (X2 ? (d & 1) == 0 : WLO)>(a[d].h, a[d].l);      // (x2c: only even k-steps carry a record)
#ifdef H3D_EXPERIMENT_TRACE_FINE
                H3D_TRACE(60 + d);                    // after it
#endif
                __builtin_amdgcn_sched_barrier(0);
                kstep(IC<1>{}, IC<1>{}, IC<(d & 1)>{}, a[d], b[d & 1], a[(d + D - 1) % D], ks + d + D - 1, b[(d + 1) & 1], ks + d + 1);
            });
        }
        // the last D k-steps: one more request, then the ring drains
        static_for<0, D>([&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            // in flight behind tail k-step d: k-steps d + 1 .. min(d + 2, D - 1) of the tail (x2c: odd ones NA loads, even ones 2 NA)
            constexpr int kTail = !X2 ? (d == 0 ? D - 2 : D - 1 - d) * NLA : (d <= 1 ? 3 * NA : d == 2 ? NA : 0);
            x3t_wait_frags<kTail, NA, (X2 ? (d & 1) == 0 : WLO)>(a[d].h, a[d].l);
            __builtin_amdgcn_sched_barrier(0);
            kstep(IC<(d == 0)>{}, IC<(d < D - 1)>{}, IC<(d & 1)>{}, a[d], b[d & 1], a[(d + D - 1) % D], KS - 1, b[(d + 1) & 1], KS - D + d + 1);
        });
    }
}

// Epilogue of one unit: y = f(rg, values of register group rg) for the four register groups of the accumulator tile
// (rg -> features 32*nt + 8*rg + 4*h + 0..3 of this lane's sample), split into hi / lo and written as the B fragments of
// k-steps 2*nt, 2*nt+1 of sample tile mt.  SPLIT(a, b, lo) -> packed hi halves.
// LO = false (precision tiers with ONE 16-bit value per activation): only the hi plane is written.
template <bool LO = true, typename SPLIT, typename F>
__device__ __forceinline__ void x3t_store_unit(const f32x16& v, unsigned char* actT, int KS, int nt, int mt, int lane, SPLIT split, F f) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rg = 2 * j + q;
            const f32x4 y = f(rg, f32x4{v[rg * 4 + 0], v[rg * 4 + 1], v[rg * 4 + 2], v[rg * 4 + 3]});
            unsigned l0, l1;
            hi[2 * q + 0] = split(y[0], y[1], l0);
            hi[2 * q + 1] = split(y[2], y[3], l1);
            lo[2 * q + 0] = l0;
            lo[2 * q + 1] = l1;
        }
        unsigned char* p = actT + x3t_frag(KS, mt, 2 * nt + j, 0) + lane * 16;
        *reinterpret_cast<u32x4*>(p) = hi;
        if constexpr (LO) *reinterpret_cast<u32x4*>(p + 1024) = lo;
    }
}

// x2 epilogue of one unit: f16 hi fragments of k-steps 2*nt, 2*nt+1 into the hi planes, and the K-tile's fp6 record (codes of
// lo * 2^12 and hi, the lane's block scale in dword 6; x3_common.hpp) split over the two "lo" planes.
// DYN: per-lane scale from the largest of the 16 values (unbounded activations); otherwise the static scale for |y| <= 1.
// gmax (DYN): running maximum of |activation| over everything this lane converts -- the x2 range guard (|x| < 2^15 keeps both
// f16 planes finite; csrc/synthesis_x3.hip: kX2ActLimit)
template <bool DYN, typename F>
__device__ __forceinline__ void x3t_store_unit_x2(const f32x16& v, unsigned char* actT, int KS, int nt, int mt, int lane, F f,
                                                  float* gmax = nullptr) {
    u32x4 hi[2], lo[2];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rg = 2 * j + q;
            const f32x4 y = f(rg, f32x4{v[rg * 4 + 0], v[rg * 4 + 1], v[rg * 4 + 2], v[rg * 4 + 3]});
            if (DYN) amax = vmax3_abs2(vmax3_abs2(amax, y[0], y[1]), y[2], y[3]);
            unsigned l0, l1;
            hi[j][2 * q + 0] = split2_x2(y[0], y[1], l0);
            hi[j][2 * q + 1] = split2_x2(y[2], y[3], l1);
            lo[j][2 * q + 0] = l0;
            lo[j][2 * q + 1] = l1;
        }
    const F16::vec8 l0v = __builtin_bit_cast(F16::vec8, lo[0]), l1v = __builtin_bit_cast(F16::vec8, lo[1]);
    const F16::vec8 h0v = __builtin_bit_cast(F16::vec8, hi[0]), h1v = __builtin_bit_cast(F16::vec8, hi[1]);
    i32x8 rec = DYN ? x2_record_dyn(l0v, l1v, h0v, h1v, amax) : x2_record(l0v, l1v, h0v, h1v);
    if (DYN && gmax) *gmax = vmax(*gmax, amax);
    rec[7] = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned char* p = actT + x3t_frag(KS, mt, 2 * nt + j, 0) + lane * 16;
        *reinterpret_cast<u32x4*>(p) = hi[j];
        *reinterpret_cast<u32x4*>(p + 1024) = u32x4{(unsigned)rec[4 * j], (unsigned)rec[4 * j + 1], (unsigned)rec[4 * j + 2], (unsigned)rec[4 * j + 3]};
    }
}

struct SplitF16 {
    __device__ __forceinline__ unsigned operator()(float a, float b, unsigned& lo) const {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const h2 hv = __builtin_convertvector(f2{a, b}, h2);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a - (float)hv.x, b - (float)hv.y}, h2));
        return __builtin_bit_cast(unsigned, hv);
    }
};
struct SplitBF16 {
    __device__ __forceinline__ unsigned operator()(float a, float b, unsigned& lo) const { return split2_bf16(a, b, lo); }
};

// value of element e of a fragment word pair (hi + lo), as fp32
__device__ __forceinline__ float x3t_f16_sum(unsigned hw, unsigned lw, int half) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, hw), b = __builtin_bit_cast(h2, lw);
    return half ? (float)a.y + (float)b.y : (float)a.x + (float)b.x;
}
__device__ __forceinline__ float x3t_bf16_sum(unsigned hw, unsigned lw, int half) {
    const unsigned a = half ? (hw & 0xffff0000u) : (hw << 16), b = half ? (lw & 0xffff0000u) : (lw << 16);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// ---- host-side packing (shared by field_x3t.hip; the synthesis side packs with torch, same layout) ------------------

inline uint16_t x3t_f32_to_f16_rn(float f) {            // round-to-nearest-even, handles subnormals; inputs are finite
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7bffu);
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - (int)(x >> 23);
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

inline float x3t_f16_to_f32(uint16_t v) {
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

// W [n_out, ld] row-major, K range [in_begin, in_begin + in_count) -> A fragments in the x3 format, [k-step][hi | lo][64][8] f16
// (scaled), KSm k-steps starting at byte `phase_off` of every tile; tiles are `tile_stride` bytes apart.
inline void x3t_pack_f16(const float* w, int ld, int in_begin, int in_count, int n_out, int NT, int64_t tile_stride, int64_t phase_off, int KSm,
                         float scale, unsigned char* dst8, bool acc_order) {
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KSm; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int k = acc_order ? x3t_acc_k(ks, lane >> 5, e) : 16 * ks + 8 * (lane >> 5) + e;
                    const int nn = 32 * nt + (lane & 31);
                    float v = 0.f;
                    if (k < in_count && nn < n_out) v = w[(int64_t)nn * ld + in_begin + k] * scale;
                    const uint16_t hi = x3t_f32_to_f16_rn(v);
                    const uint16_t lo = x3t_f32_to_f16_rn(v - x3t_f16_to_f32(hi));
                    uint16_t* dst = reinterpret_cast<uint16_t*>(dst8 + (int64_t)nt * tile_stride + phase_off + (int64_t)ks * 2048);
                    dst[lane * 8 + e] = hi;
                    dst[64 * 8 + lane * 8 + e] = lo;
                }
}

// ---- x2 packing (host): fp6 (e2m3) codes and block scales -----------------------------------------------------------
inline unsigned x2_e2m3_code(float v) {            // round-to-nearest-even on the code grid, saturating at 7.5
    const unsigned sign = v < 0.f ? 32u : 0u;
    const float a = fminf(fabsf(v), 7.5f);
    const float step = a < 2.f ? 0.125f : a < 4.f ? 0.25f : 0.5f;
    const float q = nearbyintf(a / step) * step;   // default rounding mode: ties to even; spacing doubles exactly at 2 and 4
    unsigned c;
    if (q < 2.f) c = (unsigned)(q * 8.f);
    else if (q < 4.f) c = 16u + (unsigned)((q - 2.f) * 4.f);
    else c = 24u + (unsigned)((q - 4.f) * 2.f);
    return sign | c;
}

// 32 B record of one lane and K-tile from its 16 (already scaled) weights' f16 hi values and fp32 residuals
inline void x2_make_record(const float (&hi)[16], const float (&lo)[16], unsigned (&rec)[8]) {
    float mx = 0.f;
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fabsf(hi[i]));
    int ea = mx > 0.f ? (int)floorf(log2f(7.5f / mx)) : 0;
    if (ea > 100) ea = 100;
    if (ea < -100) ea = -100;
    for (bool sat = true; sat && ea > -100;) {          // several steps when the hi values are f16 subnormals (|lo| up to |hi| / 2)
        sat = false;
        for (int i = 0; i < 16; ++i) sat = sat || fabsf(lo[i]) * kX2Rho * ldexpf(1.f, ea) > 7.5f;
        if (sat) --ea;
    }
    const float alpha = ldexpf(1.f, ea);
    for (int d = 0; d < 8; ++d) rec[d] = 0;
    for (int sl = 0; sl < 32; ++sl) {
        const float v = sl < 16 ? hi[sl] * alpha : lo[sl - 16] * alpha * kX2Rho;
        const uint64_t code = x2_e2m3_code(v);
        const int bit = 6 * sl;
        rec[bit / 32] |= (unsigned)(code << (bit & 31));
        if ((bit & 31) > 26) rec[bit / 32 + 1] |= (unsigned)(code >> (32 - (bit & 31)));
    }
    rec[6] = rec[7] = (unsigned)(127 - ea) * 0x01010101u;
}

// x3t_pack_f16 for the x2 tier, x2c format (see x3t_tile_bytes): accumulator-order matrices only, KSm (even) k-steps starting at
// byte `phase_off` (a K-tile boundary) of every tile: per K-tile [hi fragment 2T | lo record | hi fragment 2T + 1], 3 KiB.
inline void x3t_pack_x2(const float* w, int ld, int in_begin, int in_count, int n_out, int NT, int64_t tile_stride, int64_t phase_off, int KSm,
                        float scale, unsigned char* dst) {
    for (int nt = 0; nt < NT; ++nt)
        for (int T = 0; T < KSm / 2; ++T) {
            unsigned char* kt = dst + (int64_t)nt * tile_stride + phase_off + (int64_t)T * 3072;
            for (int lane = 0; lane < 64; ++lane) {
                const int nn = 32 * nt + (lane & 31), hh = lane >> 5;
                float hi[16], lo[16];
                for (int j = 0; j < 2; ++j)
                    for (int e = 0; e < 8; ++e) {
                        const int k = x3t_acc_k(2 * T + j, hh, e);
                        float v = 0.f;
                        if (k < in_count && nn < n_out) v = w[(int64_t)nn * ld + in_begin + k] * scale;
                        const uint16_t h16 = x3t_f32_to_f16_rn(v);
                        hi[8 * j + e] = x3t_f16_to_f32(h16);
                        lo[8 * j + e] = v - hi[8 * j + e];
                        reinterpret_cast<uint16_t*>(kt + j * 2048)[lane * 8 + e] = h16;
                    }
                unsigned rec[8];
                x2_make_record(hi, lo, rec);
                // lo record: code dwords 3-5 (slots 16-31) and the scale dword; the hi codes (dwords 0-2) are what
                // v_cvt_scalef32_pk32_fp6_f16 makes of the hi fragments with that scale: not stored
                unsigned* cd = reinterpret_cast<unsigned*>(kt + 1024);
                for (int d = 0; d < 3; ++d) cd[lane * 4 + d] = rec[3 + d];
                cd[lane * 4 + 3] = rec[6] & 0xffu;                   // the scale byte alone (the matrix instruction reads byte 0, the conversion shifts it)
            }
        }
}

}  // namespace h3d
