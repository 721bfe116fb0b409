// Shared pieces of the per-point MLP kernels (neural field, synthesis): packed-weight layout, fp32 MFMA
// tile engine, accurate sine.  gfx950 only.
//
// Tile engine
// -----------
// A workgroup is 256 threads = 4 wavefronts and owns a tile of 64 points (two 32-row MFMA tiles).  The
// activations of the tile live in LDS transposed, actT[k][m] with row stride MS = 68 floats, i.e. for a fixed
// input feature k the 64 points are contiguous: a v_mfma_f32_32x32x2_f32 A-operand (lane l: row l&31, k-slot
// l>>5) is then one conflict-free ds_read_b32, and an accumulator column (lane l: feature n0 + (l&31), rows
// 8*(r>>2) + 4*(l>>5) + (r&3)) is written back as four conflict-free ds_write_b128 per 32x32 tile (the +4 float
// pad rotates consecutive features over all 32 banks).
// The N (output feature) dimension is split over the four waves in 32-column tiles (wave w owns tiles w, w+4,
// ...), so every wave needs *different* weights and LDS staging of the weights would buy no reuse.  Instead the
// host packs each weight matrix once into MFMA B-fragment order
//       packed[nt][kb][lane][e] = W[k = 8*kb + 4*(lane>>5) + e][n = 32*nt + (lane&31)]
// so that a wave fetches the B operands of four consecutive MFMA k-steps for one tile with a single fully
// coalesced global_load_dwordx4 (1 KiB per wave-instruction, L2-resident), double-buffered in registers one
// k-block (8 k) ahead.  MFMA k-step e of block kb therefore contracts k = 8*kb + e (lanes 0-31) and
// k = 8*kb + 4 + e (lanes 32-63); A is read with the same mapping.
#pragma once
#include "common.hpp"

namespace h3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTileM = 64;        // points per workgroup tile
constexpr int kMS = 68;           // LDS row stride of actT (floats)
constexpr int kFieldThreads = 256;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// sin(x): Cody-Waite reduction by pi with a three-term split constant (products exact through FMA), then an odd
// Taylor polynomial on [-pi/2, pi/2].  Absolute error < 2e-7 for |x| < 1e5 (the FiLM arguments are O(10..100));
// branch-free so the unrolled epilogues stay small.
__device__ __forceinline__ float sin_accurate(float x) {
    const float k = rintf(x * 0.31830988618379067f);
    float r = fmaf(-k, 3.140625f, x);                  // pi = 3.140625 + 9.67502594e-4 + 1.50995799e-7 (+ ...)
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-7f, r);
    const float r2 = r * r;
    float p = -7.6471637318198165e-13f;                // -1/15!
    p = fmaf(p, r2, 1.6059043836821613e-10f);          //  1/13!
    p = fmaf(p, r2, -2.5052108385441720e-8f);          // -1/11!
    p = fmaf(p, r2, 2.7557319223985893e-6f);           //  1/9!
    p = fmaf(p, r2, -1.9841269841269841e-4f);          // -1/7!
    p = fmaf(p, r2, 8.3333333333333332e-3f);           //  1/5!
    p = fmaf(p, r2, -1.6666666666666666e-1f);          // -1/3!
    const float s = fmaf(r * r2, p, r);
    const int ki = (int)k;
    return __int_as_float(__float_as_int(s) ^ ((ki & 1) << 31));
}

// sin(x) through the hardware v_sin_f32 (argument in revolutions) after an exact three-constant reduction of x
// to [-pi, pi]: 7 instructions, |err| < 4e-7 on |x| < 300 (measured, tools/probes/sin_probe.hip) -- the raw
// v_sin_f32(x / 2pi) alone loses the low bits of large arguments (5e-6 at |x| = 60).
__device__ __forceinline__ float sin_hw(float x) {
    const float k = rintf(x * 0.15915494309189535f);
    float r = fmaf(-k, 6.28125f, x);                   // 2*pi = 6.28125 + 1.93500519e-3 + 3.01991598e-7 (+ ...)
    r = fmaf(-k, 1.93500518798828125e-3f, r);
    r = fmaf(-k, 3.019915981956752864e-7f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}

// One GEMM phase of the tile engine: acc[mt][i] += A(64 x 8*KB, from LDS) * W(8*KB x 32 per tile, packed).
//   aT      : LDS pointer to actT row of k = 0 for this phase
//   Wp      : packed weights of this layer ([NT][KBtot][64] float4), kb0 = first k-block to use, KBtot = blocks
//             stored per tile
// Tiles with index >= NT are clamped to tile 0 (computed and discarded) to keep the MFMA stream branch-free.
template <int NTW>
struct Frag {           // operands of one k-block (8 k): A for both 32-row tiles, B for NTW column tiles
    float a[2][4];
    float4 b[NTW];
};

template <int NTW, bool AFF>
__device__ __forceinline__ void load_frag(Frag<NTW>& f, const float* ap, const float4* const (&bp)[NTW], int kb,
                                          const float* ab, int abs, int h) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) f.b[i] = bp[i][(int64_t)kb * 64];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.a[0][e] = ap[(kb * 8 + e) * kMS];
        f.a[1][e] = ap[(kb * 8 + e) * kMS + 32];
    }
    if (AFF) {
        const float4 sc = *reinterpret_cast<const float4*>(ab + kb * 8 + 4 * h);
        const float4 sh = *reinterpret_cast<const float4*>(ab + abs + kb * 8 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = e == 0 ? sc.x : e == 1 ? sc.y : e == 2 ? sc.z : sc.w;
            const float o = e == 0 ? sh.x : e == 1 ? sh.y : e == 2 ? sh.z : sh.w;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float v = fmaf(f.a[mt][e], s, o);
                f.a[mt][e] = fmaxf(v, 0.2f * v);
            }
        }
    }
}

template <int NTW>
__device__ __forceinline__ void mfma_frag(f32x16 (&acc)[2][NTW], const Frag<NTW>& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const float bv = e == 0 ? f.b[i].x : e == 1 ? f.b[i].y : e == 2 ? f.b[i].z : f.b[i].w;
            acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][e], bv, acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][e], bv, acc[1][i], 0, 0, 0);
        }
    }
}

// The k loop is a hand-rolled two-stage software pipeline (ping-pong fragment sets): the loads of k-block kb+1
// are issued, pinned by sched_barrier, *before* the 8*NTW MFMAs (>= 1024 cycles) of k-block kb, so L2 / LDS
// latency hides under the matrix pipe.  (Left to itself hipcc sinks the prefetch back to its use and the loop
// stalls for a full L2 round trip per k-block.)
//   AFF     : the A operand is transformed on the fly, a = lrelu_0.2(x * ab[k] + ab[abs + k]) (per-input-channel
//             affine + leaky ReLU: eval-mode BatchNorm folded with a per-sample SPADE modulation); ab in LDS.
template <int NTW, bool AFF = false>
__device__ __forceinline__ void gemm_phase(f32x16 (&acc)[2][NTW], const float* aT, const float4* __restrict__ Wp,
                                           int KB, int kb0, int KBtot, int NT, int wave, int lane,
                                           const float* ab = nullptr, int abs = 0) {
    const int row = lane & 31, h = lane >> 5;
    const float4* bp[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        int nt = wave + 4 * i;
        nt = nt < NT ? nt : 0;
        bp[i] = Wp + ((int64_t)nt * KBtot + kb0) * 64 + lane;
    }
    const float* ap = aT + (4 * h) * kMS + row;
    Frag<NTW> f0, f1;
    load_frag<NTW, AFF>(f0, ap, bp, 0, ab, abs, h);
    // KB is even by construction (odd K ranges are zero padded by the packer): no conditional inside the
    // loop, otherwise LLVM sinks the prefetch into the branch that consumes it.
    for (int kb = 0; kb < KB; kb += 2) {
        load_frag<NTW, AFF>(f1, ap, bp, kb + 1, ab, abs, h);
        __builtin_amdgcn_sched_barrier(0);
        mfma_frag<NTW>(acc, f0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = kb + 2 < KB ? kb + 2 : kb;      // the last prefetch re-reads a valid block and is dropped
        load_frag<NTW, AFF>(f0, ap, bp, k2, ab, abs, h);
        __builtin_amdgcn_sched_barrier(0);
        mfma_frag<NTW>(acc, f1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NTW>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NTW]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][i][r] = 0.f;
}

// Write apply(acc, consts) back into actT.  prep(n) gathers the per-feature constants of this lane's output
// feature n once per 32x32 tile; apply(value, consts) is evaluated per element.  Features n >= n_valid are
// stored as 0 (they only ever multiply zero-padded weight rows, but must stay finite).
template <int NTW, typename P, typename F>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[2][NTW], float* actT, int NT, int n_valid, int wave,
                                          int lane, P prep, F apply) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + 4 * i;
        if (nt >= NT) continue;
        const int n = nt * 32 + j;
        const bool ok = n < n_valid;
        const auto c = prep(ok ? n : 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float4 v;
                v.x = ok ? apply(acc[mt][i][rg * 4 + 0], c) : 0.f;
                v.y = ok ? apply(acc[mt][i][rg * 4 + 1], c) : 0.f;
                v.z = ok ? apply(acc[mt][i][rg * 4 + 2], c) : 0.f;
                v.w = ok ? apply(acc[mt][i][rg * 4 + 3], c) : 0.f;
                *reinterpret_cast<float4*>(actT + n * kMS + mt * 32 + rg * 8 + 4 * h) = v;
            }
        }
    }
}

// Host-side packing of one weight matrix given in the reference's [out, in] row-major layout.
//   in_begin .. in_begin + in_count : slice of input features that forms this K range (zero padded to 8*KB)
//   dst: [NT][KB][64][4] floats
inline void pack_matrix(const float* w, int ld_in, int in_begin, int in_count, int n_out, int KB, int NT, float* dst) {
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int k = 8 * kb + 4 * (lane >> 5) + e;
                    const int n = 32 * nt + (lane & 31);
                    float v = 0.f;
                    if (k < in_count && n < n_out) v = w[(int64_t)n * ld_in + in_begin + k];
                    dst[(((int64_t)nt * KB + kb) * 64 + lane) * 4 + e] = v;
                }
}

}  // namespace h3d
