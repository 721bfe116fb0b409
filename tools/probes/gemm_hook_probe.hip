// How much VALU work per tile-pair section (6 MFMAs) does the x3 GEMM loop hide?  Synthetic hook: NF independent-chain
// v_fma_f32 + NS v_sin_f32 + NC v_cvt_pk per section, on registers (no LDS tables).  -DNF=.. -DNS=.. -DNC=..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "x3_common.hpp"
using namespace h3d;
typedef F16::vec8 half8;
#ifndef NF
#define NF 0
#endif
#ifndef NS
#define NS 0
#endif
#ifndef NR
#define NR 0
#endif
#ifndef VPM
#define VPM 4
#endif

template <int NT, int L>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* stream, int total_stages, int gemms, float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WeightRing<NT> ring;
    ring.init(stream, lds, total_stages, wave, lane);
    f32x16 acc[NT], src[NT];
    half8 xh[2 * NT], xl[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        xh[i] = half8{(_Float16)(lane * 0.001f)};
        xl[i] = half8{(_Float16)(lane * 0.0001f)};
    }
    zero_acc1<NT>(acc);
    zero_acc1<NT>(src);
    pin_agpr<NT>(src);
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed * (i + lane);
#pragma unroll 1
    for (int g = 0; g < gemms; ++g) {
        gemm_x3_roll<F16, NT, 2 * NT, 2 * NT, false, L, VPM>(acc, xh, xl, ring, [&](auto gc) __attribute__((always_inline)) {
            constexpr int gg = decltype(gc)::value;
#pragma unroll
            for (int i = 0; i < NR; ++i) r[i % 8] += src[(gg / 8) % NT][(gg * 2 + i) % 16];      // v_accvgpr_read + add
#pragma unroll
            for (int i = 0; i < NF; ++i) r[i % 8] = fmaf(r[i % 8], 1.0001f, 0.5f);
#pragma unroll
            for (int i = 0; i < NS; ++i) r[i % 8] = __builtin_amdgcn_sinf(r[i % 8]);
        });
    }
    ring.drain();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int NT = 8, KS = 16;
    const int gemms = argc > 1 ? atoi(argv[1]) : 64;
    const int wgs = argc > 2 ? atoi(argv[2]) : 256 * 8;
    const int total = 8 * KS;
    const size_t bytes = (size_t)total * NT * 2048;
    unsigned char* d; float* o;
    hipMalloc(&d, bytes); hipMemset(d, 0, bytes); hipMalloc(&o, 4096);
    const size_t lds = (size_t)H3D_RING_DEPTH * NT * 2048;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<8, 2>), dim3(wgs), dim3(256), lds, 0, d, total, gemms, o, 0.001f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double ksteps = (double)gemms * KS * (wgs / 256.0);
    printf("NR=%d NF=%d NS=%d VPM=%d: us/k-step %.4f\n", NR, NF, NS, VPM, best * 1e3 / ksteps);
    return 0;
}
