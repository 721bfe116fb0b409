// Weight-gradient GEMM of the training path for gfx950:  dW[Co, Ci] = sum_r dY[r, Co]^T X[r, Ci]  with r over the M rows
// (samples / pixels) of the batch -- M ~ 0.5 M, Co, Ci <= a few hundred.  The library GEMM handles this "tall-skinny TN" shape
// at ~50 TFLOP/s in fp32; here it is a split-K kernel on the bf16 matrix cores with split operands (x = hi + lo,
// hi*hi + hi*lo + lo*hi, fp32 accumulate: 16 mantissa bits per operand, fp32 exponent range, so gradients need no scaling).
//
//   workgroup (4 waves, 2 x 2)   owns one K-slice (rows_per_wg rows) of one [64 NA x 64 NB] block of dW; wave (wy, wx) owns
//                                NA x NB 32x32 accumulator tiles (NA = NB = 4: the whole 256 x 256 gradient in 256 registers)
//   k-step = 16 rows             the [16 x 64 NA] slab of dY and the [16 x 64 NB] slab of X are loaded once per workgroup with
//                                16-byte coalesced loads (one k-step ahead, in registers) and parked in LDS as fp32, double
//                                buffered, one barrier per k-step; a lane builds its MFMA fragments (8 consecutive rows of one
//                                column -- the contraction runs over the SLOW index of both operands) from eight ds_read_b32
//                                and splits them into bf16 hi / lo words in registers
//   output                       partial[slice][Co][Ci] fp32; the caller sums the slices (deterministic)
// HBM traffic: M (Co + Ci) 4 bytes read once per column block -- the kernel is HBM-bound at 256 x 256 (arithmetic intensity
// 3 x 2 x 256 x 256 / (512 x 4) = 192 bf16 FLOP per byte against 2.5 PFLOP/s / 5 TB/s = 500).
#include "x3_common.hpp"

namespace {

using h3d::BF16;
using h3d::f32x16;

constexpr int kThreads = 256;
constexpr int kKS = 16;                       // rows per k-step

struct Args {
    const void* dY;               // fp32, or f16 when `half` (AMP: activations and their gradients travel as f16; an f16 value's bf16
    const void* X;                // hi / lo split is exact, so the arithmetic below is unchanged and exact in the operands)
    float* partial;
    int64_t M;
    int Co, Ci, ldy, ldx, rows_per_wg;
    int conv_k, H, W, slices;     // conv_k > 0: weight gradient of a k x k convolution over [B, H, W] pixels (rows of X shifted per tap)
    float* colsum;                // optional [slices][Co]: column sums of dY over the slice's rows (the bias gradient), or null
    int half;                     // operands are _Float16 (row strides stay in elements): picks the HALF instantiation
};

// Four consecutive operand elements, as loaded: a float4 (fp32) or two words of packed f16 (AMP tier).  The prefetch keeps them
// in this form; they become fp32 only when parked in LDS, AFTER the step's MFMAs -- a conversion next to the load would make
// every load wait for its own data (eight serial HBM round trips per k-step; measured: the f16 kernels ran 2 x slower than fp32).
template <bool HALF> struct Raw4;
template <> struct Raw4<false> {
    typedef float4 type;
    static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ type load(const void* base, int64_t elem) {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
    }
    static __device__ __forceinline__ float4 widen(const type& r) { return r; }
};
template <> struct Raw4<true> {
    typedef unsigned type __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ type zero() { return type{0u, 0u}; }
    static __device__ __forceinline__ type load(const void* base, int64_t elem) {
        return *reinterpret_cast<const type*>(static_cast<const _Float16*>(base) + elem);
    }
    static __device__ __forceinline__ float4 widen(const type& r) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned w0 = r[0], w1 = r[1];          // copies first: bit_cast of a vector-element lvalue reads element 0
        const h2 a = __builtin_bit_cast(h2, w0), b = __builtin_bit_cast(h2, w1);
        return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
    }
};

// row of X that pairs with row rr of dY for filter tap (ty, tx): the shifted pixel, or -1 outside the image (zero padding)
__device__ __forceinline__ int64_t shifted_row(int64_t rr, int ty, int tx, int H, int W) {
    const int64_t HW = (int64_t)H * W;
    const int rem = (int)(rr % HW);
    const int y = rem / W, x = rem - y * W;
    if ((unsigned)(y + ty) >= (unsigned)H || (unsigned)(x + tx) >= (unsigned)W) return -1;
    return rr + (int64_t)ty * W + tx;
}

template <int NT, bool HALF, bool SHIFT = false>
__device__ __forceinline__ void load_slab(typename Raw4<HALF>::type (&r)[NT], const void* __restrict__ src, int ld, int64_t row0, int64_t M,
                                          int col0, int ncols, int t, int ty = 0, int tx = 0, int H = 1, int W = 1) {
    constexpr int W4 = 16 * NT;               // float4 per slab row (slab width 64 NT floats)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int idx = j * kThreads + t;
        const int row = idx / W4, c = (idx - row * W4) * 4;
        int64_t rr = row0 + row;
        bool ok = rr < M && col0 + c < ncols;
        if (SHIFT && ok) {
            rr = shifted_row(rr, ty, tx, H, W);
            ok = rr >= 0;
        }
        r[j] = ok ? Raw4<HALF>::load(src, rr * ld + col0 + c) : Raw4<HALF>::zero();
    }
}

// x-fused conv variant: the X band of one filter row ty for 16 output pixels = the 18 source rows r0 - 1 + ty W .. r0 + 16 + ty W
// (tx = -1, 0, +1 read rows e, e + 1, e + 2 of it).  W % 16 == 0 and r0 % 16 == 0, so the 16 pixels lie in one image row y:
// the band is zero when y + ty leaves the image, its first row is zero when the pixels start an image row (x = 0 has no left
// neighbour; that band row is only ever read as the tx = -1 neighbour of pixel 0) and its last row when they end one.
template <int NT, bool HALF>
__device__ __forceinline__ void load_band(typename Raw4<HALF>::type (&r)[NT + (NT + 7) / 8], const void* __restrict__ src, int ld, int64_t r0, int64_t r_end,
                                          int col0, int ncols, int t, int ty, int H, int W) {
    constexpr int W4 = 16 * NT;               // float4 per band row
    constexpr int NJ = NT + (NT + 7) / 8;     // 18 rows = 16 + 2: ceil(18 * W4 / 256) float4 per thread
    const int64_t HW = (int64_t)H * W;
    const int rem = (int)(r0 % HW);
    const int y = rem / W, x0 = rem - y * W;
    const bool band_ok = r0 < r_end && (unsigned)(y + ty) < (unsigned)H;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = j * kThreads + t;
        const int row = idx / W4, c = (idx - row * W4) * 4;          // band row 0..17 <-> source pixel r0 - 1 + row (+ ty W)
        bool ok = band_ok && row < 18 && col0 + c < ncols;
        if (row == 0 && x0 == 0) ok = false;
        if (row == 17 && x0 + 16 == W) ok = false;
        // pixels of this k-step past the end of the slice contribute nothing through dY (zero rows); their X rows are in range
        const int64_t q = r0 - 1 + row + (int64_t)ty * W;
        r[j] = ok ? Raw4<HALF>::load(src, q * ld + col0 + c) : Raw4<HALF>::zero();
    }
}

// LDS slabs hold the operands as they will be multiplied: fp32 (split into bf16 hi / lo by the consumer) or, in the HALF
// instantiations, the f16 values themselves -- an f16 operand is exact in one plane, so the AMP tier's weight gradient is ONE
// v_mfma_f32_32x32x16_f16 per tile and k-step instead of three bf16 products, with no conversion anywhere (round 4).
template <bool HALF> struct SlabElem { typedef float type; };
template <> struct SlabElem<true> { typedef _Float16 type; };

template <bool HALF>
__device__ __forceinline__ void park4(typename SlabElem<HALF>::type* dst, const typename Raw4<HALF>::type& r) {
    if constexpr (HALF) *reinterpret_cast<typename Raw4<true>::type*>(dst) = r;          // four halves: one 8-byte store
    else *reinterpret_cast<float4*>(dst) = r;
}

template <int NT, bool HALF>
__device__ __forceinline__ void park_band(const typename Raw4<HALF>::type (&r)[NT + (NT + 7) / 8], typename SlabElem<HALF>::type* lds, int t) {
    constexpr int W4 = 16 * NT, NJ = NT + (NT + 7) / 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = j * kThreads + t;
        if (idx < 18 * W4) park4<HALF>(lds + idx * 4, r[j]);
    }
}

template <int NT, bool HALF>
__device__ __forceinline__ void park_slab(const typename Raw4<HALF>::type (&r)[NT], typename SlabElem<HALF>::type* lds, int t) {
#pragma unroll
    for (int j = 0; j < NT; ++j) park4<HALF>(lds + (j * kThreads + t) * 4, r[j]);
}

// bf16 hi / lo fragments of column `col` of an LDS slab of width W: element e <-> row 8 * (lane >> 5) + e
template <int W>
__device__ __forceinline__ void read_frag(const float* lds, int col, int lane, BF16::vec8& hi, BF16::vec8& lo, int row0 = 0) {
    const float* p = lds + (row0 + 8 * (lane >> 5)) * W + col;
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = h3d::split2_bf16(p[(2 * e) * W], p[(2 * e + 1) * W], l[e]);
    hi = __builtin_bit_cast(BF16::vec8, h3d::u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(BF16::vec8, h3d::u32x4{l[0], l[1], l[2], l[3]});
}
// the f16 fragment of the same column: eight 16-bit LDS reads, no arithmetic
template <int W>
__device__ __forceinline__ void read_frag(const _Float16* lds, int col, int lane, h3d::F16::vec8& v, int row0 = 0) {
    const _Float16* p = lds + (row0 + 8 * (lane >> 5)) * W + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p[e * W];
}

// one k-step of a tile: three split-bf16 products, or the one exact f16 product
template <bool HALF> struct Frag {
    BF16::vec8 h, l;
    template <int W> __device__ __forceinline__ void read(const float* lds, int col, int lane, int row0 = 0) { read_frag<W>(lds, col, lane, h, l, row0); }
    static __device__ __forceinline__ f32x16 mul(const Frag& a, const Frag& b, f32x16 acc) {
        acc = BF16::mfma(a.l, b.h, acc);
        acc = BF16::mfma(a.h, b.l, acc);
        return BF16::mfma(a.h, b.h, acc);
    }
};
template <> struct Frag<true> {
    h3d::F16::vec8 v;
    template <int W> __device__ __forceinline__ void read(const _Float16* lds, int col, int lane, int row0 = 0) { read_frag<W>(lds, col, lane, v, row0); }
    static __device__ __forceinline__ f32x16 mul(const Frag& a, const Frag& b, f32x16 acc) { return h3d::F16::mfma(a.v, b.v, acc); }
};

template <int NA, int NB, bool CONV, bool HALF>
__global__ __launch_bounds__(kThreads) void wgrad_x3_kernel(Args A) {
    typedef typename Raw4<HALF>::type raw_t;
    constexpr int WA = 64 * NA, WB = 64 * NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kBuf = kKS * (WA + WB);        // floats per buffer: the dY slab, then the X slab
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int co0 = blockIdx.y * WA, ci0 = blockIdx.z * WB;
    const int tap = CONV ? blockIdx.x / A.slices : 0, slice = CONV ? blockIdx.x - tap * A.slices : blockIdx.x;
    const int ty = CONV ? tap / A.conv_k - A.conv_k / 2 : 0, tx = CONV ? tap - (tap / A.conv_k) * A.conv_k - A.conv_k / 2 : 0;
    const int64_t r_begin = (int64_t)slice * A.rows_per_wg;
    const int64_t r_end = r_begin + A.rows_per_wg < A.M ? r_begin + A.rows_per_wg : A.M;
    const int n_steps = r_end > r_begin ? (int)((r_end - r_begin + kKS - 1) / kKS) : 0;

    f32x16 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    raw_t ra[NA], rb[NB];
    // bias gradient riding along (first column block of X, first tap only): a thread's slab positions (row, 4 columns) are the
    // same in every k-step, so it keeps NA running float4 sums; the 16 rows are folded through LDS at the end
    const bool do_colsum = A.colsum != nullptr && blockIdx.z == 0 && tap == 0;
    float4 cs[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_colsum = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const float4 v = Raw4<HALF>::widen(ra[j]);
            cs[j].x += v.x; cs[j].y += v.y; cs[j].z += v.z; cs[j].w += v.w;
        }
    };      // called where the slab is parked (its loads have landed by then), never next to the loads
    // rows past r_end must not leak into this slice: the loaders clip at min(M, r_end) through the `M` argument
    load_slab<NA, HALF>(ra, A.dY, A.ldy, r_begin, r_end, co0, A.Co, t);
    load_slab<NB, HALF, CONV>(rb, A.X, A.ldx, r_begin, r_end, ci0, A.Ci, t, ty, tx, A.H, A.W);
    if (do_colsum) add_colsum();
    typedef typename SlabElem<HALF>::type slab_t;
    slab_t* slab = reinterpret_cast<slab_t*>(smem);
    park_slab<NA, HALF>(ra, slab, t);
    park_slab<NB, HALF>(rb, slab + kKS * WA, t);
    __syncthreads();

    for (int s = 0; s < n_steps; ++s) {
        const slab_t* curA = slab + (s & 1) * kBuf;
        const slab_t* curB = curA + kKS * WA;
        slab_t* nxtA = slab + ((s + 1) & 1) * kBuf;
        if (s + 1 < n_steps) {
            const int64_t row0 = r_begin + (int64_t)(s + 1) * kKS;
            load_slab<NA, HALF>(ra, A.dY, A.ldy, row0, r_end, co0, A.Co, t);
            load_slab<NB, HALF, CONV>(rb, A.X, A.ldx, row0, r_end, ci0, A.Ci, t, ty, tx, A.H, A.W);
        }
        Frag<HALF> fa[NA], fb[NB];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[a].template read<WA>(curA, (wy * NA + a) * 32 + (lane & 31), lane);
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b].template read<WB>(curB, (wx * NB + b) * 32 + (lane & 31), lane);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = Frag<HALF>::mul(fa[a], fb[b], acc[a][b]);
        if (s + 1 < n_steps) {
            if (do_colsum) add_colsum();
            park_slab<NA, HALF>(ra, nxtA, t);
            park_slab<NB, HALF>(rb, nxtA + kKS * WA, t);
        }
        __syncthreads();
    }

    if (do_colsum) {            // fold the 16 slab rows: thread (row, c) parks its sums, the first WA threads add the rows up
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<float4*>(smem + (j * kThreads + t) * 4) = cs[j];
        __syncthreads();
        if (t < WA && co0 + t < A.Co) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < kKS; ++r) v += smem[r * WA + t];
            A.colsum[(int64_t)slice * A.Co + co0 + t] = v;
        }
    }
    // accumulator tile (a, b): lane holds column ci = lane & 31, rows co = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)
    float* out = A.partial + (int64_t)blockIdx.x * A.Co * A.Ci;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ci = ci0 + (wx * NB + b) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + (wy * NA + a) * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
                if (co < A.Co && ci < A.Ci) out[(int64_t)co * A.Ci + ci] = acc[a][b][i];
            }
        }
}

// 3x3 convolution weight gradient with the three taps of a filter row fused: a workgroup owns a K-slice of pixels, one filter
// row ty and a [64 NA x 64 NB] block of dW for tx = -1, 0, +1 (3 NA NB accumulator tiles per wave, NA, NB <= 2), so dY and X are
// read three times (once per ty) instead of nine.  partial[3 ty + tx][slice][Co][Ci].
#ifndef H3D_WGRAD3_OCC
#define H3D_WGRAD3_OCC 2
#endif
template <int NA, int NB, bool HALF>
__global__ __launch_bounds__(kThreads, (HALF && NA == 2 && NB == 2) ? H3D_WGRAD3_OCC : 1) void wgrad_conv3_kernel(Args A) {
    typedef typename Raw4<HALF>::type raw_t;
    constexpr int WA = 64 * NA, WB = 64 * NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kBuf = kKS * WA + 18 * WB;     // floats per buffer: the dY slab (16 rows), then the X band (18 rows)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int co0 = blockIdx.y * WA, ci0 = blockIdx.z * WB;
    const int tyi = blockIdx.x / A.slices, slice = blockIdx.x - tyi * A.slices;
    const int ty = tyi - 1;
    const int64_t r_begin = (int64_t)slice * A.rows_per_wg;
    const int64_t r_end = r_begin + A.rows_per_wg < A.M ? r_begin + A.rows_per_wg : A.M;
    const int n_steps = r_end > r_begin ? (int)((r_end - r_begin + kKS - 1) / kKS) : 0;

    f32x16 acc[3][NA][NB];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[x][a][b][i] = 0.f;

    raw_t ra[NA], rb[NB + (NB + 7) / 8];
    // the convolution's bias gradient (column sums of dY) rides along, as in wgrad_x3_kernel: first filter row, first X block only
    const bool do_colsum = A.colsum != nullptr && blockIdx.z == 0 && tyi == 0;
    float4 cs[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_colsum = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const float4 v = Raw4<HALF>::widen(ra[j]);
            cs[j].x += v.x; cs[j].y += v.y; cs[j].z += v.z; cs[j].w += v.w;
        }
    };
    load_slab<NA, HALF>(ra, A.dY, A.ldy, r_begin, r_end, co0, A.Co, t);
    load_band<NB, HALF>(rb, A.X, A.ldx, r_begin, r_end, ci0, A.Ci, t, ty, A.H, A.W);
    if (do_colsum) add_colsum();
    typedef typename SlabElem<HALF>::type slab_t;
    slab_t* slab = reinterpret_cast<slab_t*>(smem);
    park_slab<NA, HALF>(ra, slab, t);
    park_band<NB, HALF>(rb, slab + kKS * WA, t);
    __syncthreads();

    for (int s = 0; s < n_steps; ++s) {
        const slab_t* curA = slab + (s & 1) * kBuf;
        const slab_t* curB = curA + kKS * WA;
        slab_t* nxtA = slab + ((s + 1) & 1) * kBuf;
        if (s + 1 < n_steps) {
            const int64_t row0 = r_begin + (int64_t)(s + 1) * kKS;
            load_slab<NA, HALF>(ra, A.dY, A.ldy, row0, r_end, co0, A.Co, t);
            load_band<NB, HALF>(rb, A.X, A.ldx, row0, r_end, ci0, A.Ci, t, ty, A.H, A.W);
        }
        Frag<HALF> fa[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[a].template read<WA>(curA, (wy * NA + a) * 32 + (lane & 31), lane);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            Frag<HALF> fb[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b].template read<WB>(curB, (wx * NB + b) * 32 + (lane & 31), lane, x);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[x][a][b] = Frag<HALF>::mul(fa[a], fb[b], acc[x][a][b]);
        }
        if (s + 1 < n_steps) {
            if (do_colsum) add_colsum();
            park_slab<NA, HALF>(ra, nxtA, t);
            park_band<NB, HALF>(rb, nxtA + kKS * WA, t);
        }
        __syncthreads();
    }
    if (do_colsum) {            // fold the 16 slab rows through LDS (the buffers are free now)
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<float4*>(smem + (j * kThreads + t) * 4) = cs[j];
        __syncthreads();
        if (t < WA && co0 + t < A.Co) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < kKS; ++r) v += smem[r * WA + t];
            A.colsum[(int64_t)slice * A.Co + co0 + t] = v;
        }
    }
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        float* out = A.partial + ((int64_t)(tyi * 3 + x) * A.slices + slice) * A.Co * A.Ci;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int ci = ci0 + (wx * NB + b) * 32 + (lane & 31);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co = co0 + (wy * NA + a) * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
                    if (co < A.Co && ci < A.Ci) out[(int64_t)co * A.Ci + ci] = acc[x][a][b][i];
                }
            }
    }
}

template <int NA, int NB>
int launch_conv3(const Args& a, hipStream_t st) {
    constexpr size_t lds = 2 * (kKS * 64 * NA + 18 * 64 * NB) * sizeof(float);
    const dim3 grid((unsigned)(a.slices * 3), (unsigned)((a.Co + 64 * NA - 1) / (64 * NA)), (unsigned)((a.Ci + 64 * NB - 1) / (64 * NB)));
    h3d::pre_launch();
    if (a.half) hipLaunchKernelGGL((wgrad_conv3_kernel<NA, NB, true>), grid, dim3(kThreads), lds, st, a);
    else hipLaunchKernelGGL((wgrad_conv3_kernel<NA, NB, false>), grid, dim3(kThreads), lds, st, a);
    return h3d::launch_status("h3d_conv_wgrad_x3");
}

int tiles_for(int c) { return c > 128 ? 4 : c > 64 ? 2 : 1; }

template <int NA, int NB>
int launch(const Args& a, int slices, hipStream_t st) {
    constexpr size_t lds = 2 * kKS * (64 * NA + 64 * NB) * sizeof(float);
    const int taps = a.conv_k > 0 ? a.conv_k * a.conv_k : 1;
    const dim3 grid((unsigned)(slices * taps), (unsigned)((a.Co + 64 * NA - 1) / (64 * NA)), (unsigned)((a.Ci + 64 * NB - 1) / (64 * NB)));
    h3d::pre_launch();
    if (a.conv_k > 0) {
        if (a.half) hipLaunchKernelGGL((wgrad_x3_kernel<NA, NB, true, true>), grid, dim3(kThreads), lds, st, a);
        else hipLaunchKernelGGL((wgrad_x3_kernel<NA, NB, true, false>), grid, dim3(kThreads), lds, st, a);
    } else {
        if (a.half) hipLaunchKernelGGL((wgrad_x3_kernel<NA, NB, false, true>), grid, dim3(kThreads), lds, st, a);
        else hipLaunchKernelGGL((wgrad_x3_kernel<NA, NB, false, false>), grid, dim3(kThreads), lds, st, a);
    }
    return h3d::launch_status(a.conv_k > 0 ? "h3d_conv_wgrad_x3" : "h3d_wgrad_x3");
}

}  // namespace

// Number of K-slices (= leading dimension of `partial`) and rows per slice for a problem: about one workgroup per compute unit (the 4 x 4 kernel holds 392 registers per lane: one workgroup per CU),
// at least 256 rows each.
extern "C" int h3d_wgrad_x3_slices(int64_t M, int Co, int Ci) {
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    const int na = tiles_for(Co), nb = tiles_for(Ci);
    const int64_t blocks = (int64_t)((Co + 64 * na - 1) / (64 * na)) * ((Ci + 64 * nb - 1) / (64 * nb));
    // FLOOR: one workgroup per compute unit at a time (392 registers per lane), so slices x blocks must not EXCEED the unit count --
    // rounding up (round 3) gave e.g. 86 x 3 = 258 workgroups on 256 units: a second round for two stragglers
    int64_t want = (int64_t)h3d::compute_units() / blocks;
    const int64_t most = (M + 255) / 256;
    if (want > most) want = most;
    if (want < 1) want = 1;
    return (int)want;
}

static int wgrad_x3_any(const void* dY, const void* X, int half, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                        int ldx, int slices, h3d_stream_t stream);
extern "C" int h3d_wgrad_x3_bias(const float* dY, const float* X, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                                 int ldx, int slices, h3d_stream_t stream) {
    return wgrad_x3_any(dY, X, 0, partial, colsum, M, Co, Ci, ldy, ldx, slices, stream);
}
/* h3d_wgrad_x3_bias on f16 operands (AMP, round 4): dY, X are _Float16 (row strides in elements), the gradients fp32. */
extern "C" int h3d_wgrad_x3_bias_f16(const void* dY, const void* X, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                                     int ldx, int slices, h3d_stream_t stream) {
    return wgrad_x3_any(dY, X, 1, partial, colsum, M, Co, Ci, ldy, ldx, slices, stream);
}

extern "C" int h3d_wgrad_x3(const float* dY, const float* X, float* partial, int64_t M, int Co, int Ci, int ldy, int ldx,
                            int slices, h3d_stream_t stream) {
    return h3d_wgrad_x3_bias(dY, X, partial, nullptr, M, Co, Ci, ldy, ldx, slices, stream);
}

// h3d_wgrad_x3 that also writes colsum[slice][Co] = the column sums of dY over the slice's rows (the bias gradient of the same
// layer; the caller sums the slices): dY is streamed anyway, so the separate reduction pass over it disappears.
static int wgrad_x3_any(const void* dY, const void* X, int half, float* partial, float* colsum, int64_t M, int Co, int Ci, int ldy,
                        int ldx, int slices, h3d_stream_t stream) {
    H3D_REQUIRE(dY && X && partial, "h3d_wgrad_x3: null pointer");
    H3D_REQUIRE(M >= 1 && Co >= 1 && Ci >= 1, "h3d_wgrad_x3: bad shape M=%lld Co=%d Ci=%d", (long long)M, Co, Ci);
    H3D_REQUIRE(Co % 4 == 0 && Ci % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldy >= Co && ldx >= Ci,
                "h3d_wgrad_x3: Co, Ci and the leading dimensions must be multiples of 4 (Co=%d Ci=%d ldy=%d ldx=%d)", Co, Ci, ldy, ldx);
    H3D_REQUIRE(h3d::aligned16(dY) && h3d::aligned16(X), "h3d_wgrad_x3: operands must be 16-byte aligned");
    H3D_REQUIRE(slices >= 1 && slices <= 65535 * 16, "h3d_wgrad_x3: slices=%d out of range", slices);
    Args a{};
    a.dY = dY; a.X = X; a.partial = partial; a.M = M; a.Co = Co; a.Ci = Ci; a.ldy = ldy; a.ldx = ldx; a.slices = slices;
    a.colsum = colsum; a.half = half;
    const int64_t per = (M + slices - 1) / slices;
    a.rows_per_wg = (int)(((per + kKS - 1) / kKS) * kKS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int na = tiles_for(Co), nb = tiles_for(Ci);
#define H3D_CASE(NA, NB) if (na == NA && nb == NB) return launch<NA, NB>(a, slices, st)
    H3D_CASE(4, 4); H3D_CASE(4, 2); H3D_CASE(4, 1); H3D_CASE(2, 4); H3D_CASE(2, 2); H3D_CASE(2, 1);
    H3D_CASE(1, 4); H3D_CASE(1, 2); H3D_CASE(1, 1);
#undef H3D_CASE
    h3d::set_error("h3d_wgrad_x3: no kernel for tile counts %d x %d", na, nb);
    return H3D_EUNSUPPORTED;
}

// 1 when h3d_conv_wgrad_x3 runs a 3x3 problem on the x-fused kernel (three passes over the operands instead of nine): image rows
// that are multiples of 16 pixels and enough pixels for the operand traffic to matter (the low-resolution layers' operands fit
// the caches, and their 256 x 256 blocks do more work per byte).
extern "C" int h3d_conv_wgrad_x3_fused(int B, int H, int W, int Co, int Ci) {
    (void)Co; (void)Ci;
    return W % 16 == 0 && (int64_t)B * H * W >= 32768;
}

// K-slices for h3d_conv_wgrad_x3: about two workgroups per compute unit over (slices x filter rows or taps x output blocks),
// at least 256 pixels per slice.
extern "C" int h3d_conv_wgrad_x3_slices(int B, int H, int W, int Co, int Ci, int k) {
    const int64_t M = (int64_t)B * H * W;
    if (M <= 0 || Co <= 0 || Ci <= 0) return 0;
    int64_t per_slice_wgs;
    if (k == 3 && h3d_conv_wgrad_x3_fused(B, H, W, Co, Ci)) {
        per_slice_wgs = 3 * (int64_t)((Co + 127) / 128) * ((Ci + 127) / 128);
    } else {
        const int na = tiles_for(Co), nb = tiles_for(Ci);
        per_slice_wgs = (int64_t)k * k * ((Co + 64 * na - 1) / (64 * na)) * ((Ci + 64 * nb - 1) / (64 * nb));
    }
    // FLOOR, for the same reason as in h3d_wgrad_x3_slices: 171 slices x 3 filter rows = 513 workgroups ran as THREE rounds on 256
    // units (one workgroup per unit at a time) instead of two -- a third of every fused 3x3 weight gradient's time
    int64_t want = 2 * (int64_t)h3d::compute_units() / per_slice_wgs;
    const int64_t most = (M + 255) / 256;
    if (want > most) want = most;
    if (want < 1) want = 1;
    return (int)want;
}

// Weight gradient of a k x k convolution (stride 1, zero padding k/2) of channels-last activations:
//   partial[tap][slice][Co][Ci] = sum_{p in slice} dY[p, Co]^T X[p + tap, Ci]      (the caller sums the slices)
// dY [B*H*W, Co], X [B*H*W, Ci] fp32 with row strides ldy, ldx (channel slices of wider tensors); Co, Ci multiples of 4;
// slices: any >= 1.
static int conv_wgrad_any(const void* dY, const void* X, int half, float* partial, float* colsum, int B, int H, int W, int Co, int Ci,
                          int k, int ldy, int ldx, int slices, h3d_stream_t stream);
/* ... with the bias gradient: colsum [slices][Co] receives the column sums of dY over each slice's pixels (the caller sums the
 * slices), from the pass that streams dY anyway; half = 1: f16 operands. */
extern "C" int h3d_conv_wgrad_x3_bias(const void* dY, const void* X, float* partial, float* colsum, int B, int H, int W, int Co, int Ci,
                                      int k, int ldy, int ldx, int slices, int half, h3d_stream_t stream) {
    H3D_REQUIRE(colsum, "h3d_conv_wgrad_x3_bias: null colsum");
    return conv_wgrad_any(dY, X, half, partial, colsum, B, H, W, Co, Ci, k, ldy, ldx, slices, stream);
}
extern "C" int h3d_conv_wgrad_x3(const float* dY, const float* X, float* partial, int B, int H, int W, int Co, int Ci, int k,
                                 int ldy, int ldx, int slices, h3d_stream_t stream) {
    return conv_wgrad_any(dY, X, 0, partial, nullptr, B, H, W, Co, Ci, k, ldy, ldx, slices, stream);
}
/* h3d_conv_wgrad_x3 on f16 operands (AMP, round 4): dY, X are _Float16 (row strides in elements), the gradient fp32. */
extern "C" int h3d_conv_wgrad_x3_f16(const void* dY, const void* X, float* partial, int B, int H, int W, int Co, int Ci, int k,
                                     int ldy, int ldx, int slices, h3d_stream_t stream) {
    return conv_wgrad_any(dY, X, 1, partial, nullptr, B, H, W, Co, Ci, k, ldy, ldx, slices, stream);
}
static int conv_wgrad_any(const void* dY, const void* X, int half, float* partial, float* colsum, int B, int H, int W, int Co, int Ci,
                          int k, int ldy, int ldx, int slices, h3d_stream_t stream) {
    H3D_REQUIRE(dY && X && partial, "h3d_conv_wgrad_x3: null pointer");
    H3D_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Co >= 1 && Ci >= 1 && (k == 1 || k == 3), "h3d_conv_wgrad_x3: bad shape");
    H3D_REQUIRE(Co % 4 == 0 && Ci % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldy >= Co && ldx >= Ci,
                "h3d_conv_wgrad_x3: Co, Ci and the row strides must be multiples of 4 (Co=%d Ci=%d ldy=%d ldx=%d)", Co, Ci, ldy, ldx);
    H3D_REQUIRE(h3d::aligned16(dY) && h3d::aligned16(X), "h3d_conv_wgrad_x3: operands must be 16-byte aligned");
    H3D_REQUIRE(slices >= 1 && (int64_t)slices * k * k <= 65535 * 16, "h3d_conv_wgrad_x3: slices=%d out of range", slices);
    Args a{};
    a.dY = dY; a.X = X; a.partial = partial; a.M = (int64_t)B * H * W; a.Co = Co; a.Ci = Ci; a.ldy = ldy; a.ldx = ldx;
    a.conv_k = k; a.H = H; a.W = W; a.slices = slices; a.half = half; a.colsum = colsum;
    const int64_t per = (a.M + slices - 1) / slices;
    a.rows_per_wg = (int)(((per + kKS - 1) / kKS) * kKS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k == 3 && W % 16 == 0 && h3d_conv_wgrad_x3_fused(B, H, W, Co, Ci)) {
        // the three taps of a filter row share one pass over dY and X (blocks of <= 128 x 128 so that 3 accumulator sets fit)
        const int na = Co > 64 ? 2 : 1, nb = Ci > 64 ? 2 : 1;
        if (na == 2 && nb == 2) return launch_conv3<2, 2>(a, st);
        if (na == 2 && nb == 1) return launch_conv3<2, 1>(a, st);
        if (na == 1 && nb == 2) return launch_conv3<1, 2>(a, st);
        return launch_conv3<1, 1>(a, st);
    }
    const int na = tiles_for(Co), nb = tiles_for(Ci);
#define H3D_CASE(NA, NB) if (na == NA && nb == NB) return launch<NA, NB>(a, slices, st)
    H3D_CASE(4, 4); H3D_CASE(4, 2); H3D_CASE(4, 1); H3D_CASE(2, 4); H3D_CASE(2, 2); H3D_CASE(2, 1);
    H3D_CASE(1, 4); H3D_CASE(1, 2); H3D_CASE(1, 1);
#undef H3D_CASE
    return H3D_EUNSUPPORTED;
}

// ---------------------------------------------------------------- the slices' sum (round 6)
// dW[co][ci][tap] = sum_s partial[tap][s][co][ci] and dB[co] = sum_s colsum[s][co] in ONE launch, written in the parameter's own
// layout [Co, Ci, k, k] -- what was `partial.sum(1).view(k, k, Co, Ci).permute(2, 3, 0, 1).contiguous()` + `colsum.sum(0)` on
// torch's reduction and copy kernels (three launches and an intermediate per weight gradient, 215 weight gradients per
// config-4 iteration).  A workgroup sums 256 consecutive (co, ci) elements of one tap: four waves take every fourth slice as 16-byte
// loads, the four partial sums meet in LDS in a fixed order (deterministic, like the slices themselves).
namespace {
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ colsum,
                                                           float* __restrict__ dw, float* __restrict__ db, int taps, int slices,
                                                           int64_t n_w, int n_b) {
    __shared__ float4 part[3][64];
    const bool bias = (int)blockIdx.y == taps;                    // the extra row of workgroups sums colsum
    const int tap = bias ? 0 : (int)blockIdx.y;
    const int64_t n = bias ? n_b : n_w;
    const float* src = bias ? colsum : partial + (int64_t)tap * slices * n_w;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + lane) * 4;
    if ((int64_t)blockIdx.x * 256 >= n) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n) {
        for (int s = grp; s < slices; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * n + e);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (grp) part[grp - 1][lane] = acc;
    __syncthreads();
    if (grp || e >= n) return;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float4 v = part[g][lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (bias) { *reinterpret_cast<float4*>(db + e) = acc; return; }
    if (taps == 1) { *reinterpret_cast<float4*>(dw + e) = acc; return; }
    dw[(e + 0) * taps + tap] = acc.x;
    dw[(e + 1) * taps + tap] = acc.y;
    dw[(e + 2) * taps + tap] = acc.z;
    dw[(e + 3) * taps + tap] = acc.w;
}
}  // namespace

extern "C" int h3d_wgrad_reduce(const float* partial, const float* colsum, float* dw, float* db, int taps, int slices, int Co, int Ci,
                                h3d_stream_t stream) {
    H3D_REQUIRE(partial && dw, "h3d_wgrad_reduce: null pointer");
    H3D_REQUIRE((colsum == nullptr) == (db == nullptr), "h3d_wgrad_reduce: colsum and db come together");
    H3D_REQUIRE(taps >= 1 && taps <= 9 && slices >= 1 && Co >= 1 && Ci >= 1, "h3d_wgrad_reduce: bad shape");
    H3D_REQUIRE(Co % 4 == 0 && Ci % 4 == 0, "h3d_wgrad_reduce: Co and Ci must be multiples of 4 (got %d, %d)", Co, Ci);
    H3D_REQUIRE(h3d::aligned16(partial) && h3d::aligned16(dw) && h3d::aligned16(colsum) && h3d::aligned16(db),
                "h3d_wgrad_reduce: buffers must be 16-byte aligned");
    const int64_t n_w = (int64_t)Co * Ci;
    const dim3 grid((unsigned)((n_w + 255) / 256), (unsigned)(taps + (colsum ? 1 : 0)));
    h3d::pre_launch();
    hipLaunchKernelGGL(wgrad_reduce_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), partial, colsum, dw, db, taps, slices,
                       n_w, Co);
    return h3d::launch_status("h3d_wgrad_reduce");
}
