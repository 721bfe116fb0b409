// Does VALU work hide behind MFMAs when TWO waves share a SIMD?  The x3 GEMM loop with NT = 4 tiles (64 accumulator
// registers) and NF filler FMAs per section, launched with 1 or 2 workgroups per CU (-DOCC=1|2).  Reports time per k-step of
// ONE workgroup stream; with perfect overlap OCC=2 costs the same wall time per k-step as OCC=1 for twice the work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "x3_common.hpp"
using namespace h3d;
typedef F16::vec8 half8;
#ifndef NF
#define NF 0
#endif
#ifndef OCC
#define OCC 1
#endif
constexpr int NT = 4;

__global__ __launch_bounds__(256, OCC) void probe(const unsigned char* stream, int total_stages, int gemms, float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WeightRing<NT> ring;
    ring.init(stream, lds, total_stages, wave, lane);
    f32x16 acc[NT];
    half8 xh[2 * NT], xl[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) { xh[i] = half8{(_Float16)(lane * 0.001f)}; xl[i] = half8{(_Float16)(lane * 0.0001f)}; }
    zero_acc1<NT>(acc);
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed * (i + lane);
#pragma unroll 1
    for (int g = 0; g < gemms; ++g) {
        gemm_x3_roll<F16, NT, 2 * NT, 2 * NT, false, 2, 4>(acc, xh, xl, ring, [&](auto gc) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NF; ++i) r[i % 8] = fmaf(r[i % 8], 1.0001f, 0.5f);
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
    const int KS = 2 * NT;
    const int gemms = argc > 1 ? atoi(argv[1]) : 128;
    const int wgs = 256 * 8;
    const int total = 16 * KS;
    const size_t bytes = (size_t)total * NT * 2048;
    unsigned char* d; float* o;
    hipMalloc(&d, bytes); hipMemset(d, 0, bytes); hipMalloc(&o, 4096);
    // OCC = 1: pad the allocation past half the LDS so that ONE workgroup is resident per CU whatever the register count is
    const size_t lds = (size_t)H3D_RING_DEPTH * NT * 2048 + (OCC == 1 ? 48 * 1024 : 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 256, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), lds, 0, d, total, gemms, o, 0.001f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    // k-steps executed per CU = gemms*KS*(wgs/256); MFMA-pure time per k-step (NT=4) = 12*32 cycles = 384 cycles
    printf("OCC=%d (resident WGs/CU %d) NF=%d: %.4f us per k-step per CU-slot (MFMA-pure 0.16 us at 2.4 GHz)\n", OCC, nb, NF,
           best * 1e3 / ((double)gemms * KS * (wgs / 256.0)));
    return 0;
}
