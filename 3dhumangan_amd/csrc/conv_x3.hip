// 3x3 / 1x1 convolution (stride 1, "same" zero padding) of channels-last fp32 activations on the bf16 matrix cores with split
// ("x3") operands, for gfx950: the discriminator's convolutions (reference: lib/discriminators/unet_discriminators.py:8-72, the
// nn.Conv2d(fin, fout, 3, 1, 1) / (fin, fout, 1) layers of ResBlock) -- forward AND backward-data (the same implicit GEMM on
// the flipped, transposed weights; the host packs either).  The weight gradient is h3d_conv_wgrad_x3 (wgrad_x3.hip).
//
//   out[p, o] = bias[o] + sum_{tap, i} W[o, i, tap] * x[p + tap, i]         p = (b, y, x) row-major, x / out [B, H, W, C]
//
// Implicit GEMM with M = pixels, N = output channels, K = taps x input channels, on the machinery of the register-resident x3
// engines (x3_common.hpp): a wavefront owns 32 pixels and up to 256 output channels (accumulators in AGPRs), the weights are the
// MFMA A operand, pulled once per workgroup (4 waves = 128 pixels) from L2 into the LDS ring by LDS-DMA in consumption order
// [output block][tap][channel chunk][k-step][tile][hi|lo]; the activations are the B operand: a lane reads 8 consecutive input
// channels of its (shifted) pixel -- 32 contiguous bytes, channels-last -- one chunk (<= 128 channels) ahead of the matrix
// pipe, and splits them into bf16 hi / lo fragments in registers (fp32 exponent range: gradients need no scaling; 16 mantissa
// bits per operand: ~2e-5 against fp64).  Every product is hi*hi + hi*lo + lo*hi with fp32 accumulation.
// AMP tier (HALF instantiations): the activations ARE f16 -- one exact plane -- so the loaded words are the B fragments as they
// are (no conversion, no split) on the F16 matrix instruction, the weights travel as f16 hi + f16 lo (to max(2^-22 |W|, 2^-25);
// h3d_conv_x3_pack_f16), and a product is W_hi x + W_lo x: two matrix instructions instead of three, no vector work per element.
// Several workgroups share a CU here (two by design for 1x1, up to four at NT = 2).  That is where the weight ring's former
// write-after-read argument by distance failed (x3_common.hpp: acquire; tools/conv_determinism.py; profiles/r6_conv_ring_war_race.txt)
// -- the ring now waits for its LDS reads before every stage barrier, in every kernel.
#include "x3_common.hpp"
#include <type_traits>
#include <stdlib.h>

using namespace h3d;

namespace {

// ring stages (NT * 2 KiB each): 7 with one workgroup per CU (3x3: the matrix pipe is what binds, the deep ring keeps it fed); 4 in
// the two-workgroups-per-CU instantiations (OCC = 2: the 1x1 convolutions / dense layers, which move 2 x 4 bytes per 2 x 256 flops
// and are bound by how much memory traffic a CU keeps in flight -- a second resident workgroup loads while the first computes or
// stores; round 6, tools/moments_ab.py: 0.5 M x 256 x 256 fp32 462 -> 379 us, f16 274 -> 213 us; the 3x3 shapes lose 5-12 % there)
template <int OCC> struct RingDepth { static constexpr int value = OCC == 2 ? 4 : 7; };

struct Args {
    const void* x;                 // [P, Cin] fp32, or _Float16 in the HALF instantiations (AMP: activations travel as f16)
    const unsigned char* stream;   // packed weights
    const float* bias;             // [Cout] or null
    void* out;                     // [P, Cout], same element type as x
    int64_t P;                     // B * H * W
    int H, W, Cin, Cout, k, n_chunks, stages_per_oblk;
    int ldx, ldo;                  // row strides (elements) of x and out: >= Cin / Cout (channel slices of wider tensors)
    const void* add;               // optional [P, Cout] addend of the output's type (a residual connection), row stride lda; or null
    int lda;
    float* moments;                // optional [ceil(P / 128), 2, Cout]: per workgroup, the column sums of the output as stored and of its square
    float* partial;                // split-K (gridDim.z slices of the tap x chunk loop): [slices, P, Cout] fp32 sums WITHOUT bias / addend; or null
    int it_per_slice;
};

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// this lane's 8 consecutive channels (16 ks + 8 h ..) of its shifted pixel for the KSC k-steps of one chunk: 2 float4 per k-step
template <int KSC>
__device__ __forceinline__ void load_chunk(float4 (&raw)[2 * KSC], const float* __restrict__ src, bool valid, int h) {
#pragma unroll
    for (int s = 0; s < KSC; ++s) {
        const float4* q = reinterpret_cast<const float4*>(src + 16 * s + 8 * h);
        raw[2 * s] = valid ? q[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        raw[2 * s + 1] = valid ? q[1] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// f16 activations: the lane's 8 channels of a k-step are ONE 16-byte load
template <int KSC>
__device__ __forceinline__ void load_chunk(u32x4 (&raw)[KSC], const _Float16* __restrict__ src, bool valid, int h) {
#pragma unroll
    for (int s = 0; s < KSC; ++s)
        raw[s] = valid ? *reinterpret_cast<const u32x4*>(src + 16 * s + 8 * h) : u32x4{0u, 0u, 0u, 0u};
}
template <int KSC>
__device__ __forceinline__ void split_chunk(const float4 (&raw)[2 * KSC], BF16::vec8 (&xh)[KSC], BF16::vec8 (&xl)[KSC]) {
#pragma unroll
    for (int s = 0; s < KSC; ++s) {
        unsigned hw[4], lw[4];
        hw[0] = split2_bf16(raw[2 * s].x, raw[2 * s].y, lw[0]);
        hw[1] = split2_bf16(raw[2 * s].z, raw[2 * s].w, lw[1]);
        hw[2] = split2_bf16(raw[2 * s + 1].x, raw[2 * s + 1].y, lw[2]);
        hw[3] = split2_bf16(raw[2 * s + 1].z, raw[2 * s + 1].w, lw[3]);
        xh[s] = __builtin_bit_cast(BF16::vec8, u32x4{hw[0], hw[1], hw[2], hw[3]});
        xl[s] = __builtin_bit_cast(BF16::vec8, u32x4{lw[0], lw[1], lw[2], lw[3]});
    }
}

// MODE 0: fp32 activations, split-bf16 x3.  MODE 1: f16 activations, weights as f16 hi + lo (two products per weight).  MODE 2: f16
// activations, weights rounded to f16 ONCE -- the reference's autocast semantics (lib/trainers/base_trainer.py:50-51: autocast
// rounds the weight to 11 bits) -- one product per weight, a ring stage carries two k-steps (gemm_x3_roll PAIRK), half the stream.
template <int NT, int KSC, int MODE = 0, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void conv_x3_kernel(Args A) {
    constexpr int kDepth = RingDepth<OCC>::value;
    constexpr bool HALF = MODE != 0, ONE = MODE == 2;
    constexpr int KST = ONE ? KSC / 2 : KSC;          // ring stages per chunk
    typedef typename std::conditional<HALF, _Float16, float>::type TX;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_lds[];
    constexpr int L = NT >= 4 ? 2 : 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int oblk = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * 4 + wave) * 32 + m;
    const bool okp = p < A.P;
    const int64_t pc = okp ? p : A.P - 1;
    const int64_t HW = (int64_t)A.H * A.W;
    const int rem = (int)(pc % HW);
    const int y = rem / A.W, x = rem - y * A.W;
    const int pad = A.k / 2;
    // split-K: slice blockIdx.z owns the iterations [it0, n_iter) of the tap x chunk loop -- a contiguous piece of the weight stream
    const int it0 = A.partial ? (int)blockIdx.z * A.it_per_slice : 0;
    const int n_all = A.k * A.k * A.n_chunks;
    const int n_iter = A.partial ? (it0 + A.it_per_slice < n_all ? it0 + A.it_per_slice : n_all) : n_all;

    WeightRing<NT, kDepth> ring;
    ring.init(A.stream + ((int64_t)oblk * A.stages_per_oblk + (int64_t)it0 * KST) * (NT * 2048), ring_lds,
              A.partial ? (n_iter - it0) * KST : A.stages_per_oblk, wave, lane);

    f32x16 acc[NT];
    typename std::conditional<HALF, u32x4[KSC], float4[2 * KSC]>::type raw;
    typedef typename std::conditional<HALF, F16, BF16>::type TE;          // the engine's element type
    typename TE::vec8 xh[KSC], xl[KSC];
    auto fragments = [&]() {
        if constexpr (HALF) {
#pragma unroll
            for (int s = 0; s < KSC; ++s) xh[s] = __builtin_bit_cast(F16::vec8, raw[s]);      // the f16 words as loaded
        } else {
            split_chunk<KSC>(raw, xh, xl);
        }
    };
    auto source = [&](int it, bool& valid) -> const TX* {
        const int tap = it / A.n_chunks, chunk = it - tap * A.n_chunks;
        const int ty = tap / A.k - pad, tx = tap - (tap / A.k) * A.k - pad;
        valid = okp && (unsigned)(y + ty) < (unsigned)A.H && (unsigned)(x + tx) < (unsigned)A.W;
        const int64_t q = valid ? pc + (int64_t)ty * A.W + tx : pc;
        return static_cast<const TX*>(A.x) + q * A.ldx + chunk * (16 * KSC);
    };
    {
        bool valid;
        const TX* src = source(it0, valid);
        load_chunk<KSC>(raw, src, valid, h);
    }
    // first chunk: fresh accumulators
    fragments();
    if (it0 + 1 < n_iter) {
        bool valid;
        const TX* src = source(it0 + 1, valid);
        load_chunk<KSC>(raw, src, valid, h);
    }
    gemm_x3_roll<TE, NT, KST, KSC, false, L, 0, true, !HALF, ONE>(acc, xh, xl, ring);
#pragma unroll 1
    for (int it = it0 + 1; it < n_iter; ++it) {
        pin_agpr<NT>(acc);
        fragments();
        if (it + 1 < n_iter) {
            bool valid;
            const TX* src = source(it + 1, valid);
            load_chunk<KSC>(raw, src, valid, h);
        }
        gemm_x3_roll<TE, NT, KST, KSC, false, L, 0, false, !HALF, ONE>(acc, xh, xl, ring);
    }
    ring.drain();
    // Batch moments of the output (round 6; the BatchNorm statistic of the SPADE that reads this layer, map3d_layers.py:176-190) from
    // the accumulators instead of a pass of its own over the stored tensor: every tile goes through the LDS the ring no longer needs
    // ([channel quad][pixel][4], 4 KiB per wave), a lane sums ONE channel over 16 pixels (rotated by the channel quad: 32 lanes, 32
    // banks), the halves meet through one cross-lane move, the four waves through LDS; one [2, NT * 32] row per workgroup, summed in
    // float64 by the caller (h3d_rows_sum_f64).  Pixels past the end contribute zeros.
    if (A.partial) {               // split-K: this slice's fp32 sums as they are; bias, addend and the output's type belong to the reduction
        if (okp) {
            float* __restrict__ o = A.partial + ((int64_t)blockIdx.z * A.P + p) * A.Cout + oblk * (NT * 32);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (oblk * (NT * 32) + nt * 32 >= A.Cout) break;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<float4*>(o + nt * 32 + rg * 8 + 4 * h) =
                        make_float4(acc[nt][4 * rg], acc[nt][4 * rg + 1], acc[nt][4 * rg + 2], acc[nt][4 * rg + 3]);
            }
        }
        return;
    }
    const bool mom = A.moments != nullptr;
    float* const stage = reinterpret_cast<float*>(ring_lds) + wave * 1024;
    float* const wsum = reinterpret_cast<float*>(ring_lds) + 4096;            // [4 waves][NT * 32][2]
    if (mom) __syncthreads();                                                 // the other waves' last fragment reads of the ring
    // accumulator tile nt: lane holds pixel m, channels 32 nt + 8 rg + 4 h + {0..3} in registers 4 rg .. 4 rg + 3
    if (okp || mom) {
        TX* __restrict__ o = static_cast<TX*>(A.out) + p * A.ldo + oblk * (NT * 32);
        const TX* __restrict__ ad = (A.add && okp) ? static_cast<const TX*>(A.add) + p * A.lda + oblk * (NT * 32) : nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (oblk * (NT * 32) + nt * 32 >= A.Cout) break;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = nt * 32 + rg * 8 + 4 * h;
                float4 v = make_float4(acc[nt][4 * rg], acc[nt][4 * rg + 1], acc[nt][4 * rg + 2], acc[nt][4 * rg + 3]);
                if (A.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(A.bias + oblk * (NT * 32) + n);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (ad) {                        // the residual connection's other branch, added in fp32 before the one rounding
                    if constexpr (HALF) {
                        const uint2 r = *reinterpret_cast<const uint2*>(ad + n);
                        const h16x2 r0 = __builtin_bit_cast(h16x2, r.x), r1 = __builtin_bit_cast(h16x2, r.y);
                        v.x += (float)r0[0]; v.y += (float)r0[1]; v.z += (float)r1[0]; v.w += (float)r1[1];
                    } else {
                        const float4 r = *reinterpret_cast<const float4*>(ad + n);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                }
                if constexpr (HALF) {            // round to f16 once, after the fp32 accumulation and the bias
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const h16x2 a = __builtin_convertvector(f2{v.x, v.y}, h16x2), c = __builtin_convertvector(f2{v.z, v.w}, h16x2);
                    if (okp) *reinterpret_cast<uint2*>(o + n) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c));
                    if (mom) v = make_float4((float)a[0], (float)a[1], (float)c[0], (float)c[1]);      // the moments of what was stored
                } else {
                    if (okp) *reinterpret_cast<float4*>(o + n) = v;
                }
                if (mom) *reinterpret_cast<float4*>(stage + ((2 * rg + h) * 32 + m) * 4) = okp ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (mom) {
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                const int cq = m >> 2;
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float val = stage[(cq * 32 + 16 * h + ((i + cq) & 15)) * 4 + (m & 3)];
                    s += val;
                    q = __builtin_fmaf(val, val, q);
                }
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (h == 0) *reinterpret_cast<float2*>(wsum + ((wave * NT + nt) * 32 + m) * 2) = make_float2(s, q);
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }
    if (mom) {
        __syncthreads();
        for (int idx = t; idx < NT * 64; idx += 256) {                       // (channel of this output block, statistic)
            if (oblk * (NT * 32) + (idx >> 1) >= A.Cout) break;
            const float tot = (wsum[idx] + wsum[NT * 64 + idx]) + (wsum[2 * NT * 64 + idx] + wsum[3 * NT * 64 + idx]);
            A.moments[((int64_t)blockIdx.x * 2 + (idx & 1)) * A.Cout + oblk * (NT * 32) + (idx >> 1)] = tot;
        }
    }
}

// out[p, c] = bias[c] + add[p, c] + sum over the slices of partial[s, p, c], in the output's type: the second half of a split-K launch
template <typename TX>
__global__ __launch_bounds__(256) void conv_splitk_reduce(const float* __restrict__ partial, const float* __restrict__ bias,
                                                          const TX* __restrict__ add, int lda, TX* __restrict__ out, int ldo, int64_t P,
                                                          int Cout, int slices) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (pixel, channel quad)
    const int cq = Cout / 4;
    if (idx >= P * cq) return;
    const int64_t p = idx / cq;
    const int c = (int)(idx - p * cq) * 4;
    float4 v = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < slices; ++sl) {
        const float4 q = *reinterpret_cast<const float4*>(partial + ((int64_t)sl * P + p) * Cout + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if constexpr (std::is_same<TX, float>::value) {
        if (add) {
            const float4 r = *reinterpret_cast<const float4*>(add + p * lda + c);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(out + p * ldo + c) = v;
    } else {
        if (add) {
            const uint2 r = *reinterpret_cast<const uint2*>(add + p * lda + c);
            const h16x2 r0 = __builtin_bit_cast(h16x2, r.x), r1 = __builtin_bit_cast(h16x2, r.y);
            v.x += (float)r0[0]; v.y += (float)r0[1]; v.z += (float)r1[0]; v.w += (float)r1[1];
        }
        typedef float f2 __attribute__((ext_vector_type(2)));
        const h16x2 a = __builtin_convertvector(f2{v.x, v.y}, h16x2), b = __builtin_convertvector(f2{v.z, v.w}, h16x2);
        *reinterpret_cast<uint2*>(out + p * ldo + c) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    }
}

// Weight stream of h3d_conv_x3 from OIHW fp32 weights, one launch: thread <-> (o, i, tap) of the convolution that will RUN
// (transposed: the backward-data convolution of w, W'[o][i][tap] = w[i][o][k*k - 1 - tap]).
__global__ void conv_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ stream, int Cout, int Cin, int kk, int NT,
                                 int KSC, int n_chunks, int transposed, int64_t total, int f16) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int tap = (int)(idx % kk);
    const int i = (int)((idx / kk) % Cin), o = (int)(idx / ((int64_t)kk * Cin));
    const float v = transposed ? w[((int64_t)i * Cout + o) * kk + (kk - 1 - tap)] : w[((int64_t)o * Cin + i) * kk + tap];
    unsigned short hb, lb;
    if (f16) {              // the AMP tier's streams: f16 hi + f16 lo (the weight to max(2^-22 |W|, 2^-25)) for the F16 matrix instruction
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        hb = __builtin_bit_cast(unsigned short, hi); lb = __builtin_bit_cast(unsigned short, lo);
    } else {
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        hb = __builtin_bit_cast(unsigned short, hi); lb = __builtin_bit_cast(unsigned short, lo);
    }
    const int ob = o / (32 * NT), nt = (o / 32) % NT, j = o & 31;
    const int chunk = i / (16 * KSC), ks = (i / 16) % KSC, h = (i >> 3) & 1, e = i & 7;
    if (f16 == 2) {         // one plane: [ob][tap][chunk][ks / 2][nt][ks % 2][64 lanes][8] -- a stage holds two k-steps
        const int64_t base = (((((int64_t)ob * kk + tap) * n_chunks + chunk) * (KSC / 2) + ks / 2) * NT + nt) * 2 + (ks & 1);
        stream[(base * 64 + 32 * h + j) * 8 + e] = hb;
        return;
    }
    // [ob][tap][chunk][ks][nt][hi|lo][64 lanes = 32 h + j][8]
    const int64_t base = (((((int64_t)ob * kk + tap) * n_chunks + chunk) * KSC + ks) * NT + nt) * 2;
    stream[(base * 64 + 32 * h + j) * 8 + e] = hb;
    stream[((base + 1) * 64 + 32 * h + j) * 8 + e] = lb;
}

// LDS of a launch: the ring, or -- with moments, at two tiles -- the staging area that takes its place after the last k-step
template <int NT, int OCC>
constexpr size_t lds_bytes() {
    constexpr size_t ring = (size_t)RingDepth<OCC>::value * NT * 2048, stage = (size_t)(4096 + 256 * NT) * 4;
    return ring > stage ? ring : stage;
}
template <int NT, int KSC, int MODE, int OCC>
int launch(const Args& A, int n_oblk, hipStream_t st, int slices) {
    H3D_ALLOW_MAX_LDS((conv_x3_kernel<NT, KSC, MODE, OCC>));
    const int64_t tiles = (A.P + 127) / 128;
    h3d::pre_launch();
    constexpr size_t lds = lds_bytes<NT, OCC>();
    hipLaunchKernelGGL((conv_x3_kernel<NT, KSC, MODE, OCC>), dim3((unsigned)tiles, (unsigned)n_oblk, (unsigned)slices), dim3(256), lds, st, A);
    return h3d::launch_status("h3d_conv_x3");
}

}  // namespace

// Tiling of a [Cout <- Cin] convolution: out[0] = tiles NT per output block (2, 4 or 8), out[1] = output blocks, out[2] = k-steps
// per channel chunk (4 or 8), out[3] = chunks.  -1 when the channel counts are not supported (both must be multiples of 64).
static int tiling_nt(int Cin, int Cout, int NT, int* out) {
    if (!out || Cin < 64 || Cout < 64 || Cin % 64 || Cout % 64) return -1;
    if (NT == 0) NT = Cout % 256 == 0 ? 8 : Cout % 128 == 0 ? 4 : 2;
    if ((NT != 2 && NT != 4 && NT != 8) || Cout % (32 * NT)) return -1;
    const int KSC = Cin % 128 == 0 ? 8 : 4;
    out[0] = NT; out[1] = Cout / (32 * NT); out[2] = KSC; out[3] = Cin / (16 * KSC);
    return 0;
}
extern "C" int h3d_conv_x3_tiling(int Cin, int Cout, int* out) { return tiling_nt(Cin, Cout, 0, out); }
// Tiles per output block for a launch over P output pixels (HOST helper, round 6): the widest blocking (fewest re-reads of the
// activations) whose grid -- ceil(P / 128) pixel tiles x output blocks -- still has a workgroup for every one of the 256 CUs, else the
// narrowest.  The discriminator's 512-channel 3x3 layers at 64 x 32 .. 8 x 4 pixels launched 8 .. 128 workgroups at NT = 8
// (tools/moments_ab.py: B = 4, 512 channels, 64 x 32: 199 -> 157 us at NT = 4; 32 x 16: 176 -> 107 us at NT = 2).  The blocking changes
// neither the order of any sum nor a single bit of the result.  0 when the channel counts are not supported.
extern "C" int h3d_conv_x3_nt_for(int Cin, int Cout, int64_t P) {
    int til[4];
    if (tiling_nt(Cin, Cout, 0, til)) return 0;
    static const int nt_max = [] { const char* e = getenv("H3D_CONV_NT_MAX"); return e ? atoi(e) : 8; }();      // A/B knob
    static const bool fill = [] { const char* e = getenv("H3D_CONV_FILL"); return !(e && e[0] == '0'); }();       // A/B knob: 0 = always the widest
    const int64_t tiles = (P + 127) / 128;
    int best = 0;
    for (int NT = 8; NT >= 2; NT >>= 1) {
        if (NT > nt_max || Cout % (32 * NT)) continue;
        if (!best) best = NT;                             // the widest
        if (!fill) break;
        best = NT;
        if (tiles * (Cout / (32 * NT)) >= 256) break;
    }
    return best;
}

// Pack OIHW fp32 weights `w` (device) into the stream h3d_conv_x3 reads (2 * Cout * Cin * k * k bf16, device).  transposed = 0:
// w is [Cout, Cin, k, k], the forward convolution; transposed = 1: w is [Cin, Cout, k, k] and the stream is that of its
// backward-data convolution (Cout <- Cin channels, flipped taps).
static int conv_pack_any(const float* w, void* stream, int Cout, int Cin, int k, int transposed, int f16, h3d_stream_t stream_, int NT = 0) {
    H3D_REQUIRE(w && stream && (k == 1 || k == 3), "h3d_conv_x3_pack: null pointer / kernel size");
    int til[4];
    if (tiling_nt(Cin, Cout, NT, til)) {
        h3d::set_error("h3d_conv_x3_pack: channel counts must be multiples of 64 and of 32 * NT (got %d -> %d, NT %d)", Cin, Cout, NT);
        return H3D_EUNSUPPORTED;
    }
    const int64_t total = (int64_t)Cout * Cin * k * k;
    h3d::pre_launch();
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), w,
                       static_cast<unsigned short*>(stream), Cout, Cin, k * k, til[0], til[2], til[3], transposed, total, f16);
    return h3d::launch_status("h3d_conv_x3_pack");
}
extern "C" int h3d_conv_x3_pack(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_) {
    return conv_pack_any(w, stream, Cout, Cin, k, transposed, 0, stream_);
}
// The stream h3d_conv_x3_f16 reads: the same layout with f16 hi + f16 lo planes (for the F16 matrix instruction; weights must lie
// in the f16 range, |w| < 65504: a convolution weight does).
extern "C" int h3d_conv_x3_pack_f16(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_) {
    return conv_pack_any(w, stream, Cout, Cin, k, transposed, 1, stream_);
}
// The stream h3d_conv_x3_f16x1 reads: ONE f16 plane (Cout * Cin * k * k halves), the weight rounded to f16 as autocast rounds it.
extern "C" int h3d_conv_x3_pack_f16x1(const float* w, void* stream, int Cout, int Cin, int k, int transposed, h3d_stream_t stream_) {
    return conv_pack_any(w, stream, Cout, Cin, k, transposed, 2, stream_);
}
// Any of the three streams (planes: 0 = bf16 hi + lo, 1 = f16 hi + lo, 2 = one f16 plane) blocked for NT tiles per output block
// (2, 4 or 8 with Cout % (32 NT) == 0; 0 = the default blocking of h3d_conv_x3_tiling): the stream h3d_conv_x3_ex reads at that NT.
extern "C" int h3d_conv_x3_pack_nt(const float* w, void* stream, int Cout, int Cin, int k, int transposed, int planes, int NT,
                                   h3d_stream_t stream_) {
    H3D_REQUIRE(planes >= 0 && planes <= 2, "h3d_conv_x3_pack_nt: planes %d (0 = bf16 pair, 1 = f16 pair, 2 = one f16 plane)", planes);
    return conv_pack_any(w, stream, Cout, Cin, k, transposed, planes, stream_, NT);
}

static int conv_x3_any(int mode, const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin,
                       int Cout, int k, int ldx, int ldo, h3d_stream_t stream_, const void* add = nullptr, int lda = 0,
                       float* moments = nullptr, int NT = 0, float* workspace = nullptr, int slices = 1);
static int conv_plan(int Cin, int Cout, int k, int NT_, int* til, bool& occ2);
extern "C" int h3d_conv_x3(const float* x, const void* stream, const float* bias, float* out, int B, int H, int W, int Cin,
                           int Cout, int k, int ldx, int ldo, h3d_stream_t stream_) {
    return conv_x3_any(0, x, stream, bias, out, B, H, W, Cin, Cout, k, ldx, ldo, stream_);
}
/* h3d_conv_x3 on f16 activations (AMP, round 4): x and out are _Float16 (row strides in elements, multiples of 8), fp32 bias, the
 * weight stream of h3d_conv_x3_pack_f16 (f16 hi + lo planes of the fp32 weights: the weight to max(2^-22 |W|, 2^-25), where autocast
 * rounds it to 11 bits); two F16 matrix products per weight (the activation is exact in one plane), fp32 accumulation, one rounding
 * to f16 at the store. */
extern "C" int h3d_conv_x3_f16(const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin,
                               int Cout, int k, int ldx, int ldo, h3d_stream_t stream_) {
    H3D_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0, "h3d_conv_x3_f16: row strides must be multiples of 8 halves (ldx=%d ldo=%d)", ldx, ldo);
    return conv_x3_any(1, x, stream, bias, out, B, H, W, Cin, Cout, k, ldx, ldo, stream_);
}
/* h3d_conv_x3_f16 with the weights in ONE f16 plane (stream of h3d_conv_x3_pack_f16x1): one F16 matrix product per weight, exactly
 * what the reference computes under float16 autocast (weight and activation rounded to f16, fp32 accumulation); half the weight
 * stream and half the matrix instructions of the two-plane tier. */
extern "C" int h3d_conv_x3_f16x1(const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin,
                                 int Cout, int k, int ldx, int ldo, h3d_stream_t stream_) {
    H3D_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0, "h3d_conv_x3_f16x1: row strides must be multiples of 8 halves (ldx=%d ldo=%d)", ldx, ldo);
    return conv_x3_any(2, x, stream, bias, out, B, H, W, Cin, Cout, k, ldx, ldo, stream_);
}
/* h3d_conv_x3 / _f16 / _f16x1 (mode 0 / 1 / 2) with a residual connection in the epilogue (round 6):
 * out = conv(x) + bias + add, `add` [B, H, W, Cout] of the output's type with row stride lda (elements; a multiple of 4 / 8 as ldo),
 * added in fp32 before the store -- the `x = h + x_in` of a skip block without a pass of its own. */
extern "C" int h3d_conv_x3_add(int mode, const void* x, const void* stream, const float* bias, const void* add, void* out, int B, int H,
                               int W, int Cin, int Cout, int k, int ldx, int ldo, int lda, h3d_stream_t stream_) {
    H3D_REQUIRE(mode >= 0 && mode <= 2, "h3d_conv_x3_add: mode %d (0 = fp32, 1 = f16 two planes, 2 = f16 one plane)", mode);
    H3D_REQUIRE(add && h3d::aligned16(add), "h3d_conv_x3_add: null or misaligned addend");
    const int gran = mode ? 8 : 4;
    H3D_REQUIRE(lda >= Cout && lda % gran == 0 && ldx % gran == 0 && ldo % gran == 0,
                "h3d_conv_x3_add: row strides must be multiples of %d (ldx=%d ldo=%d lda=%d)", gran, ldx, ldo, lda);
    return conv_x3_any(mode, x, stream, bias, out, B, H, W, Cin, Cout, k, ldx, ldo, stream_, add, lda);
}
/* The general entry (round 6): h3d_conv_x3_add with the addend optional, the blocking explicit and, optionally, the batch moments of
 * the output.  NT: tiles per output block the stream was packed for (h3d_conv_x3_pack_nt; 0 = the default blocking).  moments (or
 * null): [ceil(B*H*W / h3d_conv_x3_moment_rows()), 2, Cout] fp32 -- per workgroup of the launch, the column sums of the output AS
 * STORED (after bias, addend and, in the f16 modes, the rounding) and of its square.  Their float64 sum over the rows
 * (h3d_rows_sum_f64) is what h3d_channel_moments computes in a pass of its own over the stored tensor: the BatchNorm statistic of
 * the SPADE behind the layer (reference lib/components/map3d_layers.py:162, 176-190), taken from the accumulators. */
extern "C" int h3d_conv_x3_ex(int mode, const void* x, const void* stream, const float* bias, const void* add, void* out, float* moments,
                              float* workspace, int slices, int B, int H, int W, int Cin, int Cout, int k, int ldx, int ldo, int lda, int NT,
                              h3d_stream_t stream_) {
    H3D_REQUIRE(slices >= 0 && (slices <= 1 || (workspace && h3d::aligned16(workspace) && !moments)),
                "h3d_conv_x3_ex: %d K-slices need a workspace of slices * B*H*W * Cout floats (and return no moments)", slices);
    H3D_REQUIRE(mode >= 0 && mode <= 2, "h3d_conv_x3_ex: mode %d (0 = fp32, 1 = f16 two planes, 2 = f16 one plane)", mode);
    H3D_REQUIRE(!moments || h3d::aligned16(moments), "h3d_conv_x3_ex: misaligned moments buffer");
    H3D_REQUIRE(!add || h3d::aligned16(add), "h3d_conv_x3_ex: misaligned addend");
    const int gran = mode ? 8 : 4;
    H3D_REQUIRE((!add || (lda >= Cout && lda % gran == 0)) && ldx % gran == 0 && ldo % gran == 0,
                "h3d_conv_x3_ex: row strides must be multiples of %d (ldx=%d ldo=%d lda=%d)", gran, ldx, ldo, lda);
    return conv_x3_any(mode, x, stream, bias, out, B, H, W, Cin, Cout, k, ldx, ldo, stream_, add, add ? lda : 0, moments, NT,
                       slices > 1 ? workspace : nullptr, slices);
}
// The loop of a launch: tiling (NT_ = 0: the default blocking), whether it runs two workgroups per CU, iterations of the tap x chunk loop
static int conv_plan(int Cin, int Cout, int k, int NT_, int* til, bool& occ2) {
    if (tiling_nt(Cin, Cout, NT_, til)) return -1;
    // 1x1: two workgroups per CU on 4-k-step chunks (the stream's layout does not depend on the chunking); H3D_CONV_OCC=1: A/B knob
    static const bool occ2_ok = [] { const char* e = getenv("H3D_CONV_OCC"); return !(e && e[0] == '1'); }();
    occ2 = occ2_ok && k == 1;
    if (occ2 && til[2] == 8) { til[2] = 4; til[3] *= 2; }
    return k * k * til[3];
}
/* K-slices worth asking h3d_conv_x3_ex for (HOST helper, round 6): 1, or -- when even the narrowest blocking leaves CUs without a
 * workgroup -- the number of pieces of the tap x channel-chunk loop (at least 4 iterations each, at most 16 pieces) that brings the grid
 * to ~256 workgroups.  The discriminator's 16 x 8 and 8 x 4 layers (4 and 1 pixel tiles) run 9 x 512 / 16 = 288 k-steps in sequence per
 * workgroup otherwise.  The caller provides slices * B*H*W * Cout floats of workspace; the slices are summed in slice order by a second
 * launch (deterministic), which also adds bias and addend and rounds once to the output's type. */
extern "C" int h3d_conv_x3_slices(int Cin, int Cout, int k, int64_t P, int NT) {
    int til[4];
    bool occ2;
    const int n_iter = conv_plan(Cin, Cout, k, NT, til, occ2);
    static const bool split = [] { const char* e = getenv("H3D_CONV_SPLITK"); return !(e && e[0] == '0'); }();       // A/B knob
    if (n_iter < 16 || !split) return 1;          // a short loop gains less than the second launch costs (1x1, 512 channels: 18 -> 24 us)
    const int64_t wgs = ((P + 127) / 128) * til[1];
    if (wgs >= 160) return 1;
    int s = (int)((256 + wgs - 1) / wgs);
    if (s > n_iter / 4) s = n_iter / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}
/* Output pixels behind one row of h3d_conv_x3_ex's moments buffer (128: one workgroup). */
extern "C" int h3d_conv_x3_moment_rows(void) { return 128; }
static int conv_x3_any(int mode, const void* x, const void* stream, const float* bias, void* out, int B, int H, int W, int Cin,
                       int Cout, int k, int ldx, int ldo, h3d_stream_t stream_, const void* add, int lda, float* moments, int NT_,
                       float* workspace, int slices) {
    H3D_REQUIRE(x && stream && out, "h3d_conv_x3: null pointer");
    H3D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (k == 1 || k == 3), "h3d_conv_x3: bad shape / kernel size (1 or 3)");
    H3D_REQUIRE(h3d::aligned16(x) && h3d::aligned16(stream) && h3d::aligned16(out) && (!bias || h3d::aligned16(bias)),
                "h3d_conv_x3: operands must be 16-byte aligned");
    H3D_REQUIRE(ldx >= Cin && ldo >= Cout && ldx % 4 == 0 && ldo % 4 == 0, "h3d_conv_x3: row strides must be multiples of 4 and cover the channels (ldx=%d ldo=%d)", ldx, ldo);
    int til[4];
    bool occ2 = false;
    const int n_iter = (k == 1 || k == 3) ? conv_plan(Cin, Cout, k, NT_, til, occ2) : -1;
    if (n_iter < 0) {
        h3d::set_error("h3d_conv_x3: channel counts must be multiples of 64 and of 32 * NT (got %d -> %d, NT %d)", Cin, Cout, NT_);
        return H3D_EUNSUPPORTED;
    }
    if (B == 0) return H3D_OK;
    Args A{};
    A.x = x; A.stream = static_cast<const unsigned char*>(stream); A.bias = bias; A.out = out; A.add = add; A.lda = lda; A.moments = moments;
    A.P = (int64_t)B * H * W; A.H = H; A.W = W; A.Cin = Cin; A.Cout = Cout; A.k = k;
    A.n_chunks = til[3]; A.stages_per_oblk = k * k * til[3] * (mode == 2 ? til[2] / 2 : til[2]); A.ldx = ldx; A.ldo = ldo;
    H3D_REQUIRE((A.P + 127) / 128 < (int64_t(1) << 31), "h3d_conv_x3: too many pixels");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int NT = til[0], KSC = til[2];
    int nz = 1;
    if (workspace && slices > 1) {                    // split-K: ceil(n_iter / it_per_slice) slices actually run
        A.it_per_slice = (n_iter + slices - 1) / slices;
        nz = (n_iter + A.it_per_slice - 1) / A.it_per_slice;
        if (nz > 1) A.partial = workspace;
    }
    int rc = H3D_EUNSUPPORTED;
#define H3D_CASE(N, K, O) if (NT == N && KSC == K && occ2 == (O == 2)) rc = mode == 2 ? launch<N, K, 2, O>(A, til[1], st, nz) : mode == 1 ? launch<N, K, 1, O>(A, til[1], st, nz) : launch<N, K, 0, O>(A, til[1], st, nz)
    H3D_CASE(8, 8, 1); H3D_CASE(8, 4, 1); H3D_CASE(4, 8, 1); H3D_CASE(4, 4, 1); H3D_CASE(2, 8, 1); H3D_CASE(2, 4, 1);
    H3D_CASE(8, 4, 2); H3D_CASE(4, 4, 2); H3D_CASE(2, 4, 2);
#undef H3D_CASE
    if (rc != H3D_OK || !A.partial) return rc;
    const int64_t quads = A.P * (Cout / 4);
    h3d::pre_launch();
    if (mode == 0)
        hipLaunchKernelGGL(conv_splitk_reduce<float>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, A.partial, bias,
                           static_cast<const float*>(add), lda, static_cast<float*>(out), ldo, A.P, Cout, nz);
    else
        hipLaunchKernelGGL(conv_splitk_reduce<_Float16>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, A.partial, bias,
                           static_cast<const _Float16*>(add), lda, static_cast<_Float16*>(out), ldo, A.P, Cout, nz);
    return h3d::launch_status("h3d_conv_x3 (slice sum)");
}
