// Stand-alone timing of the x3 GEMM inner loop (weight ring + 3-product MFMA chain) without any epilogue work:
// what the loop itself sustains with one wave per SIMD.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950
//   -I3dhumangan_amd/csrc -Iinclude [-DH3D_EXPERIMENT_...] tools/probes/gemm_loop_probe.hip -o gemm_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "x3_common.hpp"
using namespace h3d;
typedef F16::vec8 half8;

template <int NT, int L>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* stream, int total_stages, int gemms, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WeightRing<NT> ring;
    ring.init(stream, lds, total_stages, wave, lane);
    f32x16 acc[NT];
    half8 xh[2 * NT], xl[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        xh[i] = half8{(_Float16)(lane * 0.001f)};
        xl[i] = half8{(_Float16)(lane * 0.0001f)};
    }
    zero_acc1<NT>(acc);
#pragma unroll 1
    for (int g = 0; g < gemms; ++g) {
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        gemm_x3_roll<F16, NT, 2 * NT, 2 * NT, false, L>(acc, xh, xl, ring);
    }
    ring.drain();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][7];
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
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<8, 2>), dim3(wgs), dim3(256), lds, 0, d, total, gemms, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ksteps = (double)gemms * KS * (wgs / 256.0);
        printf("ms %.3f  us/k-step %.4f  (768 MFMA cycles at 2.4 GHz = 0.320 us)  TF(x3 issued) %.1f\n", ms, ms * 1e3 / ksteps,
               (double)wgs * 4 * gemms * KS * NT * 3 * 32768.0 / ms / 1e9);
    }
    return 0;
}
