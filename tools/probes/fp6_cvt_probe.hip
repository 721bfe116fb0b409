// Probe of the fp6 (e2m3) conversion instructions and their pairing with v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 fp6_cvt_probe.hip -o fp6_cvt_probe ; run on a GPU box.
//   1. v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32: code of element i at bits [6i, 6i+6)?  x / scale or x * scale?
//      rounding (nearest-even) and saturation at +-7.5?
//   2. code order of the conversion == slot order of the matrix instruction (what the x2 engines rely on)
//   3. op_sel of the scale operands: which byte of the scale VGPR is used
//   4. issue cost of the conversions beside MFMAs (cycles per loop iteration from s_memtime)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));

__global__ void cvt16(const _Float16* x, unsigned* o, float s) {
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = x[threadIdx.x * 32 + i];
    const u6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, s);
    for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
__global__ void cvt32(const float* x, unsigned* o, float s) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = x[threadIdx.x * 32 + i]; b[i] = x[threadIdx.x * 32 + 16 + i]; }
    const u6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, s);
    for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
// A, B: 32 f16 per lane; both converted on the device, then D = A6 * B6 (scaled MFMA); scale VGPRs and op_sel from the host
template <int OA, int OB>
__global__ void mm(const _Float16* A, const _Float16* B, float* D, float sa, float sb, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    h32 va, vb;
    for (int i = 0; i < 32; ++i) { va[i] = A[l * 32 + i]; vb[i] = B[l * 32 + i]; }
    const u6 ra = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(va, sa), rb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vb, sb);
    const i32x8 a = {(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)ra[4], (int)ra[5], 0, 0};
    const i32x8 b = {(int)rb[0], (int)rb[1], (int)rb[2], (int)rb[3], (int)rb[4], (int)rb[5], 0, 0};
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, OA, scale_a, OB, scale_b);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = acc[r];
}

// cost: per iteration 4 f16 MFMAs (+ 2 fp6 MFMAs) (+ NC conversions of live data); cycles per iteration of wave 0
template <int NC, int NX>
__global__ __launch_bounds__(256) void cost(const _Float16* x, float* out, long long* cyc, int n) {
    const int l = threadIdx.x & 63;
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = x[l * 32 + i];
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = v[i]; b[i] = v[8 + i]; }
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    u6 r = {0, 0, 0, 0, 0, 0};
    i32x8 a6 = {1, 2, 3, 4, 5, 6, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            v[c] = (_Float16)((float)v[c] + 1.f);                     // keep the conversion inside the loop
            const u6 t = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, 0.25f);
            r = r ^ t;
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const i32x8 b6 = {(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
            acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc[q], 2, 2, 0, 0x74747474, 0, 0x7f7f7f7f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(r[0] ^ r[5]);
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float e2m3(unsigned v) {
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float r = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return s ? -r : r;
}
static unsigned q_e2m3(float x) {                  // nearest-even, saturating
    const unsigned s = x < 0 || (x == 0 && signbit(x)) ? 32 : 0;
    float a = fabsf(x);
    if (!(a == a)) return s | 31;
    if (a >= 7.5f) return s | 31;
    unsigned best = 0; float bd = 1e30f;
    for (unsigned c = 0; c < 32; ++c) {
        const float d = fabsf(e2m3(c) - a);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return s | best;
}
static unsigned code_at(const unsigned* w, int i) {
    const int bit = 6 * i;
    uint64_t v = w[bit / 32];
    if (bit / 32 + 1 < 6) v |= (uint64_t)w[bit / 32 + 1] << 32;
    return (unsigned)(v >> (bit & 31)) & 63;
}
static float h2f(_Float16 h) { return (float)h; }

int main() {
    srand(3);
    static _Float16 hx[64 * 32], hy[64 * 32];
    static float fx[64 * 32];
    for (int i = 0; i < 64 * 32; ++i) {
        const float m = ((rand() & 0xffff) / 65536.f * 2.f - 1.f);
        fx[i] = m * ldexpf(1.f, (rand() % 7) - 3);       // |x| up to 8 at scale 1: exercises saturation and subnormals
        if (i % 97 == 0) fx[i] = 0.f;
        hx[i] = (_Float16)fx[i];
        hy[i] = (_Float16)(((rand() & 0xffff) / 65536.f * 2.f - 1.f) * ldexpf(1.f, (rand() % 5) - 2));
    }
    _Float16 *dx, *dy; float *dfx, *dD; unsigned* dout; long long* dcyc;
    hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy); hipMalloc(&dfx, sizeof fx); hipMalloc(&dout, 64 * 6 * 4); hipMalloc(&dD, 64 * 16 * 4);
    hipMalloc(&dcyc, 8);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
    hipMemcpy(dfx, fx, sizeof fx, hipMemcpyHostToDevice);
    static unsigned ho[64 * 6];
    for (int which = 0; which < 2; ++which)
        for (int t = 0; t < 3; ++t) {
            const float s = t == 0 ? 1.f : t == 1 ? 0.25f : 4.f;
            if (which == 0) cvt16<<<1, 64>>>(dx, dout, s); else cvt32<<<1, 64>>>(dfx, dout, s);
            hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
            int bad_div = 0, bad_mul = 0, first = -1;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 32; ++i) {
                    const float x = which == 0 ? h2f(hx[l * 32 + i]) : fx[l * 32 + i];
                    const unsigned got = code_at(ho + l * 6, i);
                    const unsigned wd = q_e2m3(x / s), wm = q_e2m3(x * s);
                    const bool okd = got == wd || (e2m3(got) == 0 && e2m3(wd) == 0), okm = got == wm || (e2m3(got) == 0 && e2m3(wm) == 0);
                    if (!okd) { ++bad_div; if (first < 0) first = l * 32 + i; }
                    if (!okm) ++bad_mul;
                }
            printf("%s scale %-5g: code i at bits [6i,6i+6), q(x/scale): %d mismatches; q(x*scale): %d mismatches of 2048\n",
                   which == 0 ? "cvt_scalef32_pk32_fp6_f16  " : "cvt_scalef32_2xpk16_fp6_f32", s, bad_div, bad_mul);
            if (bad_div && bad_mul && first >= 0) {
                const int l = first / 32;
                printf("   first mismatching lane %d: x =", l);
                for (int i = 0; i < 32; ++i) printf(" %g", which == 0 ? h2f(hx[l * 32 + i]) : fx[l * 32 + i]);
                printf("\n   decoded:");
                for (int i = 0; i < 32; ++i) printf(" %g", e2m3(code_at(ho + l * 6, i)));
                printf("\n");
            }
        }
    // 2 + 3: conversion order == MFMA slot order; op_sel byte selection
    {
        static float hD[64 * 16];
        const int bytes_a = 0x7f | (0x7d << 8) | (0x80 << 16) | (0x79 << 24);      // exponents 0, -2, +1, -6 in bytes 0..3
        const int bytes_b = 0x7f | (0x7e << 8) | (0x82 << 16) | (0x7c << 24);      // 0, -1, +3, -3
        const int ea[4] = {0, -2, 1, -6}, eb[4] = {0, -1, 3, -3};
        for (int oa = 0; oa < 4; ++oa) {
            const int ob = (oa + 1) & 3;
            if (oa == 0) mm<0, 1><<<1, 64>>>(dx, dy, dD, 1.f, 0.5f, bytes_a, bytes_b);
            if (oa == 1) mm<1, 2><<<1, 64>>>(dx, dy, dD, 1.f, 0.5f, bytes_a, bytes_b);
            if (oa == 2) mm<2, 3><<<1, 64>>>(dx, dy, dD, 1.f, 0.5f, bytes_a, bytes_b);
            if (oa == 3) mm<3, 0><<<1, 64>>>(dx, dy, dD, 1.f, 0.5f, bytes_a, bytes_b);
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            int bad = 0; double worst = 0, ratio = 0; int nr = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                    double acc = 0;
                    for (int h = 0; h < 2; ++h)
                        for (int s = 0; s < 32; ++s)
                            acc += (double)e2m3(q_e2m3(h2f(hx[(32 * h + i) * 32 + s]))) * (double)e2m3(q_e2m3(h2f(hy[(32 * h + j) * 32 + s]) / 0.5f));
                    const double want = acc * ldexp(1.0, ea[oa] + eb[ob]);
                    const double err = fabs(want - hD[l * 16 + r]) / (fabs(want) + 1e-2);
                    if (err > 1e-5) ++bad;
                    if (err > worst) worst = err;
                    if (fabs(acc) > 1) { ratio += hD[l * 16 + r] / acc; ++nr; }
                }
            printf("cvt -> MFMA, op_sel_a %d op_sel_b %d: slot order + byte op_sel-th of the scale VGPR %s (mismatches %d / 1024, worst %.2e; mean D/unscaled = 2^%.2f, expected 2^%d)\n",
                   oa, ob, bad ? "FAILS" : "holds", bad, worst, log2(fabs(ratio / nr)), ea[oa] + eb[ob]);
        }
    }
    // 4: cost
    {
        float* dout2; hipMalloc(&dout2, 1024 * 256 * 4);
        long long c;
        const int n = 20000;
#define RUN(NC, NX)                                                                                             \
        cost<NC, NX><<<1024, 256>>>(dx, dout2, dcyc, n); hipDeviceSynchronize();                                   \
        hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);                                                            \
        printf("cost: 4 f16 MFMA + %d cvt_pk32_fp6_f16 + %d fp6 MFMA per iteration: %.1f cycles / iteration\n", NC, NX, (double)c / n);
        RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(4, 0) RUN(0, 2) RUN(1, 2) RUN(2, 2) RUN(0, 4)
    }
    return 0;
}
