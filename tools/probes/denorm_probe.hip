// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs, and does v_cvt_pk_f16_f32 produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float tiny) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)1024.f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    const half2v h = __builtin_convertvector(f32x2{tiny, tiny * 3.f}, half2v);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)h.x; out[2] = (float)h.y; out[3] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    for (float tiny : {9.5367431640625e-7f /*2^-20*/, 5.9604645e-8f /*2^-24*/, 3.0e-6f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
        float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("tiny %.4e: mfma sum %.6e (expect %.6e)  cvt_pk -> %.4e %.4e  scalar cvt %.4e\n", tiny, h[0], 16.0 * tiny * 1024.0, h[1], h[2], h[3]);
    }
    return 0;
}
