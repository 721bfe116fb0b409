// Shared machinery of the split-operand ("x3") matrix-core engines (field_x3.hip: f16, synthesis_x3.hip: bf16).
//
//   * WeightRing   workgroup-shared LDS ring filled by LDS-DMA (global_load_lds) from a linear weight stream
//   * gemm_x3      per-wave GEMM, weights from the ring, activations as register fragments, three partial
//                  products per tile (hi*hi, hi*lo, lo*hi) on v_mfma_f32_32x32x16_{f16,bf16}
//   * relayout     accumulator layout -> next layer's B fragments with v_permlane32_swap (verified on hardware by
//                  tools/probes/x3_probe.hip)
// Fragment convention (both operands): lane l holds 8 consecutive k of row/column (l & 31): k = 16*ks + 8*(l>>5) + e.
#pragma once
#include "field_common.hpp"

namespace h3d {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// LDS table reads go through this ext-vector type, NOT HIP's float4 struct: an aggregate (struct) load from LDS makes
// the compiler's waitcnt pass assume it may alias the in-flight LDS-DMA and emit s_waitcnt vmcnt(0) in front of it,
// which drains the whole weight-ring prefetch queue at every table read.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int I>
struct IC { static constexpr int value = I; };

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

#ifndef H3D_RING_DEPTH
#define H3D_RING_DEPTH 7
#endif

struct F16 {          // scaled f16 halves
    typedef _Float16 elem;
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    typedef _Float16 vec2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(const vec8& a, const vec8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
struct BF16 {         // bf16 halves, no scaling needed (fp32 exponent range)
    typedef __bf16 elem;
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    typedef __bf16 vec2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(const vec8& a, const vec8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

template <typename T>
__device__ __forceinline__ void split(float xs, typename T::elem& hi, typename T::elem& lo) {
    hi = (typename T::elem)xs;
    lo = (typename T::elem)(xs - (float)hi);
}

// Two fp32 -> packed bf16 hi halves (returned) and packed bf16 lo halves: 6 VALU ops per pair.
__device__ __forceinline__ unsigned split2_bf16(float a, float b, unsigned& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, b2));
    const float fa = __builtin_bit_cast(float, hb << 16), fb = __builtin_bit_cast(float, hb & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a - fa, b - fb}, b2));
    return hb;
}

template <typename T>
__device__ __forceinline__ unsigned pack2(typename T::elem a, typename T::elem b) {
    return __builtin_bit_cast(unsigned, typename T::vec2{a, b});
}

// Four 32-bit words (2 elems each) per register group rg = 0..3 of one 32x32 accumulator tile, hi or lo plane:
// P[rg][0..1] hold rows 8rg+4h+{0,1} and +{2,3} of this lane's column.  Returns the B fragments of the two k-steps
// the tile spans (k = feature index).  Lanes 0-31 keep rows 8rg..+3 of the even group and receive rows +4..+7 from
// lanes 32-63; lanes 32-63 receive the odd group's rows from lanes 0-31 and keep their own.
template <typename T>
__device__ __forceinline__ void relayout_tile(const unsigned (&P)[4][2], typename T::vec8& f0, typename T::vec8& f1) {
    u32x4 r[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            auto a = __builtin_amdgcn_permlane32_swap(P[2 * pr][c], P[2 * pr + 1][c], false, false);
            r[pr][c] = a[0];
            r[pr][2 + c] = a[1];
        }
    f0 = __builtin_bit_cast(typename T::vec8, r[0]);
    f1 = __builtin_bit_cast(typename T::vec8, r[1]);
}

// LDS pointers are carried as address-space-3 pointers built from integers (ring position + lane slot), so every access
// through them is a ds_read whatever the optimiser can or cannot infer about their provenance.
typedef const __attribute__((address_space(3))) unsigned char* lds_ptr;
template <typename V>
__device__ __forceinline__ V lds_ld(lds_ptr p) { return *reinterpret_cast<const __attribute__((address_space(3))) V*>(p); }

// Per-lane base of an LDS table: the table's address plus this lane's `lane_bytes`, as ONE opaque register.  Reads then address
// `base + compile-time constant`, which instruction selection folds into the DS instruction's 16-bit offset field.  Without the
// launder LLVM re-associates "table + (lane part + constant)" into a loop-invariant (lane part + constant) per read site and
// hoists it: round 3's synthesis kernel carried ~60 such registers per convolution and paid one v_add per table read.
__device__ __forceinline__ lds_ptr lane_base(const void* table, unsigned lane_bytes) {
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)table + lane_bytes;
    asm volatile("" : "+v"(a));
    return reinterpret_cast<lds_ptr>(a);
}
__device__ __forceinline__ f32x4 ldt4(lds_ptr base, int float_index) { return lds_ld<f32x4>(base + 4 * float_index); }
// The same read as a VOLATILE access: it keeps its place in program order against the sections' sched barriers and the ring's asm
// statements.  For software-prefetched table rows: instruction selection otherwise sinks the (pure) load to its first use, one
// section later, behind that section's burst of weight-fragment reads -- the wait for it is then s_waitcnt lgkmcnt(0), a drain of
// the whole look-ahead (seen in the ISA of field_x3_kernel: every FiLM table read was the YOUNGEST read at its use).
__device__ __forceinline__ f32x4 ldt4_pinned(lds_ptr base, int float_index) {
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) f32x4*>(base + 4 * float_index);
}

// Workgroup-shared weight ring in LDS, filled by LDS-DMA.  Stage = one k-step of one matrix = NT*2 chunks of 1 KB
// ([tile][hi/lo][64 lanes][16 B]); the stream is linear in memory and wraps after `total` stages.
// LAG = 1 keeps the buffer of stage t - 1 readable during stage t (the refill issued after acquire(t) is stage
// t + kBuf - 2 and lands in the buffer of stage t - 2): the x2 engines read an fp6 record that spans two consecutive stages.
//
// Bookkeeping (round 4): the ring's state is three running ADDRESSES in scalar registers -- fill position in the stream,
// fill and read position in the ring -- each advanced by one add / compare / select per stage; the DMA takes the stream
// address as an SGPR pair (saddr form: global_load_lds_dwordx4 v_lane_slot, s[base:base+1] offset:imm) and an LDS pointer is
// one v_lshl_add_u32 of the lane id onto the scalar ring position.  Every update is laundered through an empty asm volatile,
// which is ordered against the (asm volatile) DMA issues: round 3's index arithmetic (stage index * 16 KiB as a 64-bit shift,
// buffer index * 16 KiB, 64-bit vector adds per stage) was hoisted by the compiler to the head of each GEMM, sixteen stages
// at a time, and spilled from there (v_writelane: 129 spilled SGPRs in synthesis_x3_kernel<8, 4, false, true>).
#ifndef H3D_DMA_SLOT
#define H3D_DMA_SLOT 0
#endif
template <int NT, int DEPTH = H3D_RING_DEPTH, int LAG = 0>
struct WeightRing {
    static constexpr int kBuf = DEPTH;
    static_assert(DEPTH - LAG >= 3 && (DEPTH - 2 - LAG) * (NT * 2 / 4) < 64, "ring depth out of range for the 6-bit vmcnt field");
    static constexpr int kChunks = NT * 2 / 4;          // DMA instructions per wave per stage
    static constexpr int kStage = NT * 2048;
    static constexpr unsigned kRingBytes = (unsigned)kBuf * kStage;
    // all scalar (wave-uniform); the three running positions are ABSOLUTE addresses, wrapped by compare-and-select
    const unsigned char* fill_g;            // global address of the stage filled next
    const unsigned char* g_begin;           // the weight stream
    unsigned g_end_lo;                      // low word of the stream's end (the stream is far smaller than 4 GiB)
    unsigned fill_m0, m0_begin, m0_end;     // LDS address (M0) of this wave's quarter of the buffer filled next / of buffer 0 / past the last
    unsigned read_at, rd_begin, rd_end;     // LDS address of the buffer acquired next / of the ring / past it
    int vslot;                              // vector: this lane's 16 bytes inside a stage = (wave * kChunks) KiB + lane * 16
    int lane;
    int wave;                               // scalar: this wave's index in the workgroup (staggered refills)
    bool primed = false;                    // experiments only

    __device__ __forceinline__ void init(const unsigned char* stream, unsigned char* lds, int total_stages, int w, int l) {
#ifdef H3D_EXPERIMENT_SMALL_STREAM
        total_stages = total_stages < 8 ? total_stages : 8;        // timing experiment: the stream wraps inside 128 KB (always L2 hits; wrong results)
#endif
        g_begin = fill_g = stream;
        g_end_lo = (unsigned)(size_t)stream + (unsigned)total_stages * kStage;
        rd_begin = read_at = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
        rd_end = rd_begin + kRingBytes;
        m0_begin = fill_m0 = __builtin_amdgcn_readfirstlane(rd_begin + w * kChunks * 1024);
        m0_end = m0_begin + kRingBytes;
        vslot = w * kChunks * 1024 + l * 16;
        lane = l;
        wave = w;
        asm volatile("" : "+s"(fill_g), "+s"(fill_m0), "+s"(read_at));
#pragma unroll
        for (int i = 0; i < kBuf - 1 - LAG; ++i) issue();
        primed = true;
    }
    // Staggered refill (H3D_RING_STAGGER): the kChunks pieces a wave owes per stage are issued in ONE section of the k-step,
    // a different one for each wave (section c <-> wave c), instead of one piece per section by all four waves at once: the
    // waves of a workgroup run in lockstep between barriers, so their pieces otherwise arrive at the CU's vector-memory path
    // together and queue behind each other (issue cost of a piece: ~60 cycles alone, 100-185 in a crowd, MI355X_MICROARCH.md).
    template <int C>
    __device__ __forceinline__ void issue_slot() {
#ifdef H3D_RING_STAGGER
        if ((wave % kChunks) == C) issue();
#else
        issue_chunk<C>();
#endif
    }
    // One 1 KB piece of the stage being filled.  Issued through inline asm on purpose: hipcc models
    // global_load_lds as a FLAT access that touches both LDS and memory and, while one is pending, degrades EVERY
    // later s_waitcnt it inserts to vmcnt(0) / lgkmcnt(0) -- each LDS fragment read then stalls for the full LDS
    // latency and each table read drains the whole weight prefetch queue.  The ring is synchronised by hand
    // (acquire()), so the compiler does not need to know about these writes.  The global and the LDS address advance
    // by the same 1 KB per piece, so the instruction's immediate offset serves both (one address per stage).
    // M0 is written with the first piece of a stage only: the pieces of a stage are issued sections apart, but nothing the
    // compiler generates for these kernels touches M0 in between (tests/test_abi.py disassembles the library and checks that
    // every write of M0 in it is one of these).
    template <int C>
    __device__ __forceinline__ void issue_chunk() {           // C = 0 .. kChunks-1, in order
#ifdef H3D_EXPERIMENT_NO_DMA                                  // timing experiment (wrong results): the refills are never issued
        if (primed) return;
#endif
#ifndef H3D_RING_M0_PER_PIECE
#if !defined(H3D_RING_POLICY_ID) || H3D_RING_POLICY_ID == 0      // cache-policy bits of the refill (experiments: 1 nt, 2 sc0, 3 sc1)
#define H3D_RING_POLICY ""
#elif H3D_RING_POLICY_ID == 1
#define H3D_RING_POLICY " nt"
#elif H3D_RING_POLICY_ID == 2
#define H3D_RING_POLICY " sc0"
#else
#define H3D_RING_POLICY " sc1"
#endif
        if (C == 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0" H3D_RING_POLICY : : "s"(fill_m0), "v"(vslot), "s"(fill_g) : "m0");
        else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" H3D_RING_POLICY : : "v"(vslot), "s"(fill_g), "n"(C * 1024));
#else
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" : : "s"(fill_m0), "v"(vslot), "s"(fill_g), "n"(C * 1024) : "m0");
#endif
        if (C == kChunks - 1) {
            fill_g += kStage;
            fill_g = (unsigned)(size_t)fill_g == g_end_lo ? g_begin : fill_g;
            fill_m0 += kStage;
            fill_m0 = fill_m0 == m0_end ? m0_begin : fill_m0;
            asm volatile("" : "+s"(fill_g), "+s"(fill_m0));   // ordered after the DMA above: the next stage's arithmetic stays here
        }
    }
    __device__ __forceinline__ void issue() {
        static_for<0, kChunks>([&](auto c) __attribute__((always_inline)) { issue_chunk<decltype(c)::value>(); });
    }
    // scalar LDS address of the buffer acquired next; advances the read position
    __device__ __forceinline__ unsigned next_read() {
        const unsigned at = read_at;
        read_at += kStage;
        read_at = read_at == rd_end ? rd_begin : read_at;
        asm volatile("" : "+s"(read_at));
        return at;
    }
    template <int SHIFT>
    __device__ __forceinline__ lds_ptr slot(unsigned at) const {        // lane * 2^SHIFT bytes into the buffer at `at`
        // the stage's ds_reads depend on this (opaque) address: they cannot be hoisted above the barrier
        unsigned a = ((unsigned)lane << SHIFT) + at;
        asm volatile("" : "+v"(a));
        return reinterpret_cast<lds_ptr>(a);
    }
    // Make the next stage (t) readable by every wave.  The caller then issues stage t + kBuf - 1 with
    // issue_chunk(0..kChunks-1), one chunk after each tile pair's MFMAs of the k-step it computes next, so the DMA
    // issue hides under the matrix pipe.  The refill lands in the buffer of stage t - 1.
    // Write-after-read safety (round 6: BY CONSTRUCTION).  Every wave waits for its own LDS reads (lgkmcnt(0)) before it arrives
    // at the barrier, so when the barrier opens the last fragment reads of stage t - 1 have RETURNED in every wave, and the
    // refill is issued after the barrier.  Rounds 2-5 argued by distance instead (the reads were only *issued* before the barrier;
    // the refill's data lands an L2 round trip later, an LDS read retires within ~130 cycles) because the wait cost ~15 % on the
    // round-2 kernels.  On today's kernels it costs nothing (same lease, profiles/r6_ab_ring_wait_lds.txt: 358.2 / 359.3 vs
    // 360.5 / 359.7 images/s; every convolution shape within 1 %) -- and the distance argument FAILED where several workgroups
    // share a CU: conv_x3.hip at two to four workgroups per CU with another workgroup's moments epilogue loading the LDS pipe
    // returned a tile computed from a half-refilled stage in 1-4 % of launches (profiles/r6_conv_ring_war_race.txt).
    // H3D_RING_DISTANCE restores the old wait (development only).
    __device__ __forceinline__ lds_ptr acquire() {
        // vmcnt only (expcnt / lgkmcnt fields left at "no wait"): stages t+1 .. t+kBuf-2 may stay in flight
        constexpr int kKeep = (kBuf - 2 - LAG) * kChunks;
#ifndef H3D_EXPERIMENT_NO_BARRIER
#ifndef H3D_RING_DISTANCE           // every LDS read of this wave returned before the barrier
        __builtin_amdgcn_s_waitcnt((kKeep & 0xF) | ((kKeep >> 4) << 14) | 0x0070 | 0x0000);
#else
        __builtin_amdgcn_s_waitcnt((kKeep & 0xF) | ((kKeep >> 4) << 14) | 0x0070 | 0x0F00);
#endif
        __builtin_amdgcn_s_barrier();
#endif
        return slot<4>(next_read());
    }
    // Two consecutive stages (t, t + 1) behind ONE wait + barrier (the x2 GEMM consumes stages in pairs: an fp6 record spans an
    // even / odd pair): half the workgroup barriers of acquire() per stage.  One stage fewer stays in flight while waiting.  The
    // write-after-read argument of acquire() carries over: the refills issued during the two k-steps that follow land in the
    // buffers of stages t - 1 - LAG and t - LAG, whose last reads every wave issued before it arrived here.
    // r0 / r1: lane * 16 into the even / odd stage (fragments, first record half); r1c / r1s: lane * 8 and lane * 4 into the odd
    // stage (the dense second record half: code dwords 4-5 as [64 lanes][8 B], scale dwords as [64 lanes][4 B] behind them);
    // r1d = r1c + one tile (2 KiB) as a register of its own -- the odd tile of a pair reads its code dwords through it, so that
    // the two 64-bit reads of a pair have different base registers and are NOT merged into one ds_read2st64_b64 (whose four
    // result registers would then have to be copied into the two 6-register operands: 4 v_mov per pair).
    __device__ __forceinline__ void acquire2(lds_ptr& r0, lds_ptr& r1, lds_ptr& r1c, lds_ptr& r1d, lds_ptr& r1s) {
        static_assert(kBuf - 3 - LAG >= 0, "ring too shallow for paired acquires");
        constexpr int kKeep = (kBuf - 3 - LAG) * kChunks;
#ifndef H3D_EXPERIMENT_NO_BARRIER
#ifndef H3D_RING_DISTANCE
        __builtin_amdgcn_s_waitcnt((kKeep & 0xF) | ((kKeep >> 4) << 14) | 0x0070 | 0x0000);
#else
        __builtin_amdgcn_s_waitcnt((kKeep & 0xF) | ((kKeep >> 4) << 14) | 0x0070 | 0x0F00);
#endif
        __builtin_amdgcn_s_barrier();
#endif
        r0 = slot<4>(next_read());
        const unsigned odd = next_read();
        r1 = slot<4>(odd);
        r1c = slot<3>(odd);
        r1d = slot<3>(odd + 2048);
        r1s = slot<2>(odd);
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

template <typename T, bool SWAP>
__device__ __forceinline__ f32x16 mm(const typename T::vec8& w, const typename T::vec8& x, const f32x16& c) {
#ifdef H3D_EXPERIMENT_NO_MFMA
    f32x16 r = c; r[0] += (float)w[0] * (float)x[0]; return r;
#endif
    return SWAP ? T::mfma(x, w, c) : T::mfma(w, x, c);
}

// acc[nt] (+)= W(tile nt, k-step ks) x X(k-step ks) for all tiles / k-steps of one matrix, weights from the ring.
//   SWAP = false: D[feature][sample] (weights = A operand);  SWAP = true: D[sample][feature].
// Instead of double-buffering a whole k-step of weight fragments (NT*16 registers), a rolling window of L+1 tile pairs (16 registers each) runs L pairs (L * 192 MFMA cycles) ahead of
// the matrix pipe.  Same ring protocol: acquire(s+1) happens inside k-step s, just before the first read of stage
// s+1 and after the last read of stage s was issued; each acquire is followed by exactly one refill, one DMA chunk
// after each of the next NT/2 tile pairs.  acc is accumulated into (initialise it with the bias / residual).
#ifdef H3D_EXPERIMENT_TRACE
// Development: cycle trace of workgroup (1000, 3), kept in LDS while the kernel runs (a global store per event would wait on
// vmcnt and drain the weight ring's DMA queue) and copied out at the end (H3D_TRACE_DUMP).
constexpr int kTraceMax = 500;
__shared__ unsigned long long h3d_tr_buf[kTraceMax];
__shared__ int h3d_tr_cnt;
#define H3D_TRACE(tag)                                                                         \
    do {                                                                                       \
        if (blockIdx.x == 1000 && blockIdx.y == 3 && threadIdx.x == 0) {                       \
            const int n_ = h3d_tr_cnt;                                                         \
            if (n_ < kTraceMax) {                                                              \
                h3d_tr_buf[n_] = (__builtin_readcyclecounter() << 8) | (unsigned long long)(tag); \
                h3d_tr_cnt = n_ + 1;                                                           \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#define H3D_TRACE_INIT() do { if (threadIdx.x == 0) h3d_tr_cnt = 0; __syncthreads(); } while (0)
#define H3D_TRACE_RESET() do { if (threadIdx.x == 0) h3d_tr_cnt = 0; } while (0)
#define H3D_TRACE_DUMP(dst)                                                                    \
    do {                                                                                       \
        if (blockIdx.x == 1000 && blockIdx.y == 3 && threadIdx.x == 0) {                       \
            unsigned long long* d_ = reinterpret_cast<unsigned long long*>(dst);               \
            for (int i_ = 0; i_ < h3d_tr_cnt; ++i_) d_[i_] = h3d_tr_buf[i_];                   \
        }                                                                                      \
    } while (0)
#else
#define H3D_TRACE(tag) do { } while (0)
#define H3D_TRACE_INIT() do { } while (0)
#define H3D_TRACE_RESET() do { } while (0)
#define H3D_TRACE_DUMP(dst) do { } while (0)
#endif

struct NoHook {
    template <typename G> __device__ __forceinline__ void operator()(G) const {}
};

// HOOK: side work (the caller's VALU epilogue of the previous layer, producing the fragments of later k-steps),
// called as hook(IC<g>{}) once per tile-pair section g = ks * NT/2 + p, in program order before the section's six
// MFMAs; VALU_PER_MFMA > 0 asks the scheduler to slot that many VALU instructions behind each MFMA of the section.
// Every index is a compile-time constant (static_for), so fragment / accumulator arrays stay in registers.
// XLO = false: the activation operand is exact in ONE plane (f16 inputs on the F16 engine, the AMP tier): two products, W_hi x +
// W_lo x, and xl is not read.
// PAIRK = true (with XLO = false): BOTH operands are exact in one plane (the AMP tier with the weights rounded to f16 once, which
// is what autocast does to them): a stage's two planes hold the weight fragments of TWO consecutive k-steps, plane 0 x xh[2s] +
// plane 1 x xh[2s + 1] -- one product per weight, half the stream of the two-plane format; KS counts stages (= k-steps / 2).
template <typename T, int NT, int KS, int KSA, bool SWAP, int L, int VALU_PER_MFMA = 0, bool ZERO = false, bool XLO = true, bool PAIRK = false,
          typename RING, typename HOOK = NoHook>
__device__ __forceinline__ void gemm_x3_roll(f32x16 (&acc)[NT], const typename T::vec8 (&xh)[KSA],
                                             const typename T::vec8 (&xl)[KSA], RING& ring, HOOK hook = HOOK()) {
    constexpr int P = NT / 2, G = KS * P, NB = L + 1;
    static_assert(NT % 2 == 0 && KS * (PAIRK ? 2 : 1) <= KSA && L >= 1 && L <= P, "look-ahead is at most one k-step");
    static_assert(!PAIRK || !XLO, "paired k-steps: one plane per operand");
    static_assert(RING::kChunks == P, "one DMA chunk per tile pair");
    struct Pair { typename T::vec8 h[2], l[2]; } buf[NB];
    lds_ptr st[2];
    auto load_pair = [&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        const lds_ptr s = st[(q / P) & 1] + (q % P) * 4096;
        Pair& b = buf[q % NB];
#ifdef H3D_EXPERIMENT_NO_WREAD
        if (q >= NB) return;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            b.h[i] = __builtin_bit_cast(typename T::vec8, lds_ld<u32x4>(s + (i * 2 + 0) * 1024));
            b.l[i] = __builtin_bit_cast(typename T::vec8, lds_ld<u32x4>(s + (i * 2 + 1) * 1024));
        }
    };
    H3D_TRACE(1);
    st[0] = ring.acquire();
    ring.issue();
    static_for<0, L>(load_pair);
    static_for<0, G>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int s = g / P, p = g % P;
#ifdef H3D_EXPERIMENT_TRACE_FINE
        if constexpr (s >= 4 && s < 7) H3D_TRACE(10 + p);
#endif
        if constexpr (p == P - L && s + 1 < KS) {
            if constexpr (s % 4 == 0) H3D_TRACE(2);
            st[(s + 1) & 1] = ring.acquire();
            if constexpr (s % 4 == 0) H3D_TRACE(3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // refill chunk owed to the latest acquire (its place in the section: H3D_DMA_SLOT, an experiment knob; 0 = after the MFMAs)
        auto dma = [&]() __attribute__((always_inline)) {
            constexpr int since = g - (P - L);                    // sections since the first in-loop acquire position
            if constexpr (since >= 0 && since / P + 1 < KS) ring.template issue_slot<since % P>();
        };
#if H3D_DMA_SLOT == 1
        dma();
#endif
        if constexpr (g + L < G) load_pair(IC<g + L>{});
#if H3D_DMA_SLOT == 2
        dma();
#endif
#ifndef H3D_EXPERIMENT_NO_HOOK
        hook(gc);
#endif
#if H3D_DMA_SLOT == 3
        dma();
#endif
        const Pair& b = buf[g % NB];
        constexpr int n0 = 2 * p, n1 = 2 * p + 1;
        constexpr int sx = PAIRK ? 2 * s : s;    // activation fragment of plane 0 (plane 1: the same one, or the next k-step's)
        if constexpr (ZERO && s == 0) {          // fresh accumulators: C = 0 is an inline constant, no init pass
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[n0] = mm<T, SWAP>(b.h[0], xh[sx], zero);
            acc[n1] = mm<T, SWAP>(b.h[1], xh[sx], zero);
        } else {
            acc[n0] = mm<T, SWAP>(b.h[0], xh[sx], acc[n0]);
            acc[n1] = mm<T, SWAP>(b.h[1], xh[sx], acc[n1]);
        }
#if !defined(H3D_EXPERIMENT_PRODUCTS) || H3D_EXPERIMENT_PRODUCTS >= 2      // timing experiments only (wrong results)
        if constexpr (XLO) {
            acc[n0] = mm<T, SWAP>(b.h[0], xl[s], acc[n0]);
            acc[n1] = mm<T, SWAP>(b.h[1], xl[s], acc[n1]);
        }
#endif
#if !defined(H3D_EXPERIMENT_PRODUCTS) || H3D_EXPERIMENT_PRODUCTS >= 3
        acc[n0] = mm<T, SWAP>(b.l[0], xh[PAIRK ? sx + 1 : sx], acc[n0]);
        acc[n1] = mm<T, SWAP>(b.l[1], xh[PAIRK ? sx + 1 : sx], acc[n1]);
#endif
        // refill chunk owed to the latest acquire: acquires sit at section (s*P + P-L) for s+1 < KS, each followed by
        // P chunks in the next P sections; the prologue acquire was refilled by ring.issue()
#if H3D_DMA_SLOT == 0
        dma();
#endif
        if constexpr (VALU_PER_MFMA > 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, VALU_PER_MFMA, 0);
            }
        }
        // anchor the section: instruction selection orders pure instructions (MFMAs included) by register-pressure heuristics
        // only, and has been seen to sink every MFMA of a GEMM below ALL its weight-fragment reads (which then spill); the
        // empty asm consumes the section's accumulators, so the section's matrix instructions are issued before it
        asm volatile("" : "+a"(acc[n0]), "+a"(acc[n1]));
        __builtin_amdgcn_sched_barrier(0);
    });
    H3D_TRACE(4);
}

// ---------------------------------------------------------------------------------------------------------------- x2
// "x2" arithmetic: W.x = hi.hi on v_mfma_f32_32x32x16_f16 + the two cross terms (hi.lo + lo.hi) in ONE block-scaled fp6
// instruction, v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3 operands): the cross terms are 2^-11 of the main term and need only a
// few significant bits.  Per K-tile (32 input features = 2 k-steps) and output tile: 2 f16 MFMAs (64 cycles) + 1 fp6 MFMA
// (32 cycles) instead of 6 f16 MFMAs (192 cycles), and 83 nJ instead of 150 nJ (tools/probes/mfma_scale_probe.hip) -- the
// engines are power-limited, so the energy is what sets the clock.
//   operand slots of the fp6 instruction (verified by tools/probes/{mfma_scale,fp6_cvt}_probe.hip): lane half h of A
//   contracts slot s (bits [6s, 6s+6) of the lane's 6 dwords) with slot s of lane half h of B; per-lane e8m0 scales.
//   A record (weights, 32 B per lane and K-tile): slots 0-15 = q6(hi(W)) of the lane's 16 features (acc order: slot 8j+e =
//   element e of k-step 2T+j), slots 16-31 = q6(lo(W)); dword 6 = the lane's block scale (all four bytes), dword 7 = the same
//   (gemm_x2_roll reads the scale from dword 7: a separate 32-bit load next to the 64-bit load of code dwords 4-5, see there).
//   B record (activations): v_cvt_scalef32_pk32_fp6_f16 of [lo'(k-step 2T), lo'(2T+1), hi(2T), hi(2T+1)] (lo' = lo * 2^12).
//   Stream: a stage is still one k-step, [tile][1 KiB f16 hi fragment][1 KiB]; the second KiB of an x2 k-step holds dwords
//   0-3 (even k-step) / 4-7 (odd k-step) of the K-tile's A records, of an x3 k-step (inputs assembled from memory) the f16
//   lo fragment as before.  The record is read in the odd k-step from two ring buffers (WeightRing LAG = 1).
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

constexpr int kX2ScaleB = 127 - 14;        // static e8m0 of the cross terms for |x| <= 1: hi * 4, lo * 2^12 * 4  ->  2^-14 with rho = 2^12
constexpr float kX2Rho = 4096.f;           // lo planes travel multiplied by rho
constexpr float kX2CvtScale = 0.25f;       // v_cvt_scalef32 divides by its scale: codes = q6(x * 4)

// the K-tile's activation record from its four f16 fragments: codes = q6(value / cvt_scale), true value = code * 2^(scale_byte
// - 127) (hi) resp. * 2^-12 of that (lo'); dword 6 = the lane's e8m0 scale byte (op_sel 0 reads byte 0), dword 7 unused
__device__ __forceinline__ i32x8 x2_record(const F16::vec8& l0, const F16::vec8& l1, const F16::vec8& h0, const F16::vec8& h1,
                                           float cvt_scale = kX2CvtScale, int scale_byte = kX2ScaleB + 12) {
    typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
    const f16x16 lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const f16x16 hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const f16x32 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
                                             23, 24, 25, 26, 27, 28, 29, 30, 31);
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, cvt_scale);
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    i32x8 o = __builtin_bit_cast(i32x8, (u32x8)__builtin_shufflevector(r, r, 0, 1, 2, 3, 4, 5, -1, -1));   // dword 7 stays undefined
    o[6] = scale_byte - 12;                  // both cross terms carry the 2^-12 of the lo planes
    return o;
}

template <bool SWAP>
__device__ __forceinline__ f32x16 mm6(const i32x8& w, const i32x8& x, const f32x16& c) {
    const int sw = w[6], sx = x[6];          // the lanes' block scales (weights: 1 / alpha; activations: see x2_record)
#ifdef H3D_EXPERIMENT_NO_MFMA
    f32x16 r = c; r[0] += (float)(w[0] + x[0] + sw + sx); return r;
#endif
    return SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, w, c, 2, 2, 0, sx, 0, sw)
                : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, x, c, 2, 2, 0, sw, 0, sx);
}

// two fp32 -> packed f16 hi halves (returned) and packed f16 halves of lo * 2^12 (plain VALU only, see split2_act)
// max / |max| without the canonicalising v_max x, x, x pairs the compiler puts in front of every fmaxf (IEEE sNaN quieting;
// v_max_f32 quiets by itself) and without the explicit v_and of fabsf (|.| is a free source modifier): measured in the x2
// synthesis engine, lrelu + running maximum cost 7.5 VALU instructions per activation before and 2.5 after.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vrelu(float a) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ float vmax_abs2(float a, float b) {                 // max(|a|, |b|)
    float r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3_abs2(float m, float a, float b) {       // max(m, |a|, |b|)
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ unsigned split2_x2(float a, float b, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 hv = __builtin_convertvector(f2{a, b}, h2);
#ifndef H3D_X2_SPLIT_PLAIN
    // mixed-precision FMAs read the f16 halves directly and write packed f16: residual (exact), then * 2^12 and the conversion --
    // 5 instructions per pair instead of 8, bit-identical results (synthesis engine 25.6 -> 25.1 ms)
    const unsigned hw = __builtin_bit_cast(unsigned, hv);
    float la, lb;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hw), "v"(a));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hw), "v"(b));
    unsigned lw;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(lw) : "v"(la), "s"(kX2Rho));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(lw) : "v"(lb), "s"(kX2Rho));
    lo = lw;
    return hw;
#else
    const float fa = (float)hv.x, fb = (float)hv.y;
    float la, lb;
    asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(fa));
    asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(fb));
    asm("v_mul_f32 %0, %1, %2" : "=v"(la) : "v"(la), "v"(kX2Rho));
    asm("v_mul_f32 %0, %1, %2" : "=v"(lb) : "v"(lb), "v"(kX2Rho));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{la, lb}, h2));
    return __builtin_bit_cast(unsigned, hv);
#endif
}

// ... for |a|, |b| < 16 (the field: sine outputs), one instruction less per pair: hi * 2^12 is still an f16 (exact), and one
// mixed FMA per value gives f16(a * 2^12 - hi * 2^12) -- the same single rounding of the same exact residual, so the same bits.
__device__ __forceinline__ unsigned split2_x2_bounded(float a, float b, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef H3D_X2_SPLIT_GENERAL
    return split2_x2(a, b, lo);
#endif
    const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2));
    unsigned hs, lw;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(hs) : "v"(hw), "s"(0x6C006C00u));                 // (4096, 4096) as f16
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lw) : "v"(a), "s"(kX2Rho), "v"(hs));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw) : "v"(b), "s"(kX2Rho), "v"(hs));
    lo = lw;
    return hw;
}

// The fp6 record of a K-tile whose largest |activation| of this lane is amax = m * 2^e (1 <= m < 2): codes = q6(y / cs) with
// cs = 2^(e-2) when m < 1.875 (largest hi code < 7.5: no saturation) and 2^(e-1) otherwise.  lo' = lo * 2^12 <= 2^(e+1) can
// reach code 8 in the first case and is then clipped to 7.5 (an error of 2^-4 of a term that is 2^-12 of the product).
__device__ __forceinline__ i32x8 x2_record_dyn(const F16::vec8& l0, const F16::vec8& l1, const F16::vec8& h0, const F16::vec8& h1, float amax) {
    unsigned eb = (__builtin_bit_cast(unsigned, amax) + 0x100000u) >> 23;   // biased exponent, + 1 when m >= 1.875 (amax >= 0)
    eb = eb < 16u ? 16u : eb;                                               // all-zero / tiny tiles: any in-range scale
    return x2_record(l0, l1, h0, h1, __builtin_bit_cast(float, (eb - 2u) << 23), (int)eb - 2);
}

// acc[nt] (+)= W x X over KS2 x2 k-steps (KS2 even; B operands xh[s] and the K-tile records b6[s / 2]) followed by KS3 x3
// k-steps (B operands xh[s], xl3[s - KS2]: fragments assembled from memory, lo unscaled).  Same ring protocol, look-ahead
// and hook convention as gemm_x3_roll; a section carries 2 (even x2 k-step), 4 (odd) or 6 (x3) MFMAs.
// PRE: called as pre(IC<g>{}) at the top of section g, BEFORE the section's weight-fragment reads are issued: the place for a
// software prefetch (a pinned LDS read) that must be OLDER than that burst, so that the wait for it one section later is
// s_waitcnt lgkmcnt(burst size) and not a drain of the look-ahead.
template <int NT, int KS2, int KS3, int KSA, int KT, bool SWAP, int L, int VALU_PER_MFMA = 0, bool ZERO = false, typename RING, typename HOOK = NoHook,
          typename PRE = NoHook>
__device__ __forceinline__ void gemm_x2_roll(f32x16 (&acc)[NT], const F16::vec8 (&xh)[KSA], const i32x8 (&b6)[KT],
                                             const F16::vec8 (&xl3)[KS3 > 0 ? KS3 : 1], RING& ring, HOOK hook = HOOK(), PRE pre = PRE()) {
    typedef F16 T;
    constexpr int KS = KS2 + KS3, P = NT / 2, G = KS * P, NB = L + 1;
    static_assert(NT % 2 == 0 && KS2 % 2 == 0 && KS <= KSA && KS2 / 2 <= KT && L >= 1 && L <= P, "look-ahead is at most one k-step");
    static_assert(RING::kChunks == P, "one DMA chunk per tile pair");
    // record halves: dwords 0-3 (stage s - 1) and 4-6 (stage s: codes 4, 5 and the scale) -- the second as a 64-bit and a 32-bit
    // load, so that the code dwords land in sub-registers of the instruction's ONE 6-register operand (a 128-bit second half
    // cannot: the compiler then copies two dwords per fp6 instruction)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct Pair { typename T::vec8 h[2], l[2]; u32x4 c0[2]; u32x2 c1[2]; unsigned sc[2]; } buf[NB];
    lds_ptr st[2];
    lds_ptr stc[2] = {nullptr, nullptr}, sts = nullptr;   // the odd stage of the current pair at lane * 8 (even / odd tile) and lane * 4
    auto load_pair = [&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, s = q / P;
        const lds_ptr b0 = st[s & 1] + (q % P) * 4096;
        Pair& b = buf[q % NB];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            b.h[i] = __builtin_bit_cast(typename T::vec8, lds_ld<u32x4>(b0 + (i * 2 + 0) * 1024));
            if constexpr (s >= KS2) {
                b.l[i] = __builtin_bit_cast(typename T::vec8, lds_ld<u32x4>(b0 + (i * 2 + 1) * 1024));
            } else if constexpr (s % 2 == 1) {
                // record halves: code dwords 0-3 from the even stage (16 B per lane: one conflict-free ds_read_b128), code dwords
                // 4-5 and the scale dword from the odd stage, stored DENSE -- [64 lanes][8 B] then [64 lanes][4 B] -- so that the
                // 64-bit and the 32-bit read are conflict-free as well (round 3 kept 16 B per lane there: the lanes of a half-wave
                // then hit every fourth bank pair, a 2-way conflict on the b64 and a 4-way conflict on the b32 read, 1.6e9 conflict
                // cycles per launch).  Two loads, not one of 96 bits: the code dwords must land in sub-registers of the
                // instruction's ONE 6-register operand and the scale in a register of its own.
                b.c0[i] = lds_ld<u32x4>(st[(s - 1) & 1] + (q % P) * 4096 + (i * 2 + 1) * 1024);
                b.c1[i] = lds_ld<u32x2>(stc[i] + (q % P) * 4096 + 1024);
                b.sc[i] = lds_ld<unsigned>(sts + (q % P) * 4096 + (i * 2 + 1) * 1024 + 512);
            }
        }
    };
    // stages are acquired in pairs (one barrier per two k-steps): (0, 1) here, (s + 1, s + 2) inside every odd k-step s
    if constexpr (KS2 >= 2) ring.acquire2(st[0], st[1], stc[0], stc[1], sts);
    else st[0] = ring.acquire();
    ring.issue();
    static_for<0, L>(load_pair);
    static_for<0, G>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int s = g / P, p = g % P;
        if constexpr (p == P - L && s + 1 < KS) {
            if constexpr (s + 1 >= KS2) st[(s + 1) & 1] = ring.acquire();                            // x3 tail: single stages
            else if constexpr (s % 2 == 1) ring.acquire2(st[(s + 1) & 1], st[s & 1], stc[0], stc[1], sts);      // x2: the pair (s + 1, s + 2)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (p == 0) H3D_TRACE(100 + s);
        auto dma = [&]() __attribute__((always_inline)) {
            constexpr int since = g - (P - L);
            if constexpr (since >= 0 && since / P + 1 < KS) ring.template issue_slot<since % P>();
        };
#if H3D_DMA_SLOT == 1
        dma();
#endif
        pre(gc);
        if constexpr (g + L < G) load_pair(IC<g + L>{});
#if H3D_DMA_SLOT == 2
        dma();
#endif
        hook(gc);
#if H3D_DMA_SLOT == 3
        dma();
#endif
        const Pair& b = buf[g % NB];
        constexpr int n0 = 2 * p, n1 = 2 * p + 1;
        if constexpr (ZERO && s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[n0] = mm<T, SWAP>(b.h[0], xh[s], zero);
            acc[n1] = mm<T, SWAP>(b.h[1], xh[s], zero);
        } else {
            acc[n0] = mm<T, SWAP>(b.h[0], xh[s], acc[n0]);
            acc[n1] = mm<T, SWAP>(b.h[1], xh[s], acc[n1]);
        }
        constexpr int n_mfma = s >= KS2 ? 6 : s % 2 == 1 ? 4 : 2;
        if constexpr (s >= KS2) {
            acc[n0] = mm<T, SWAP>(b.h[0], xl3[s - KS2], acc[n0]);
            acc[n1] = mm<T, SWAP>(b.h[1], xl3[s - KS2], acc[n1]);
            acc[n0] = mm<T, SWAP>(b.l[0], xh[s], acc[n0]);
            acc[n1] = mm<T, SWAP>(b.l[1], xh[s], acc[n1]);
        } else if constexpr (s % 2 == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const i32x8 w6 = {(int)b.c0[i][0], (int)b.c0[i][1], (int)b.c0[i][2], (int)b.c0[i][3],
                                  (int)b.c1[i][0], (int)b.c1[i][1], (int)b.sc[i], 0};
                acc[n0 + i] = mm6<SWAP>(w6, b6[s / 2], acc[n0 + i]);
            }
        }
#if H3D_DMA_SLOT == 0
        dma();
#endif
        if constexpr (VALU_PER_MFMA > 0) {
#pragma unroll
            for (int i = 0; i < n_mfma; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, VALU_PER_MFMA, 0);
            }
        }
        asm volatile("" : "+a"(acc[n0]), "+a"(acc[n1]));          // anchor (see gemm_x3_roll)
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Force an accumulator set into the AGPR half of the register file at this point (MFMA reads / writes C there
// directly; VALU users pay one v_accvgpr_read).  Used to hand the compiler the partition "accumulators in AGPRs,
// fragments and scalars in VGPRs" that it does not find by itself at 500 live registers.
__device__ __forceinline__ void pin1(f32x16& v) { asm volatile("" : "+a"(v)); }

template <int NT>
__device__ __forceinline__ void pin_agpr(f32x16 (&v)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("" : "+a"(v[i]));
}

template <int NT>
__device__ __forceinline__ void zero_acc1(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

}  // namespace h3d
