// Accuracy of candidate device sine implementations against double precision, on the FiLM argument range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__device__ float sin_cw(float x) {       // Cody-Waite (3 constants) + Taylor-15, the shipped sin_accurate
    const float k = rintf(x * 0.31830988618379067f);
    float r = fmaf(-k, 3.140625f, x); r = fmaf(-k, 9.67502593994140625e-4f, r); r = fmaf(-k, 1.509957990978376432e-7f, r);
    const float r2 = r * r;
    float p = -7.6471637318198165e-13f; p = fmaf(p, r2, 1.6059043836821613e-10f); p = fmaf(p, r2, -2.5052108385441720e-8f);
    p = fmaf(p, r2, 2.7557319223985893e-6f); p = fmaf(p, r2, -1.9841269841269841e-4f); p = fmaf(p, r2, 8.3333333333333332e-3f);
    p = fmaf(p, r2, -1.6666666666666666e-1f);
    const float s = fmaf(r * r2, p, r);
    return __int_as_float(__float_as_int(s) ^ (((int)k & 1) << 31));
}
__device__ float sin_hw1(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }    // v_sin_f32 on x/(2pi)
__device__ float sin_hw2(float x) {       // exact reduction to [-pi, pi] first, then v_sin_f32
    const float k = rintf(x * 0.15915494309189535f);
    float r = fmaf(-k, 6.28125f, x); r = fmaf(-k, 1.93500518798828125e-3f, r); r = fmaf(-k, 3.019915981956752864e-7f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
__global__ void k(const float* x, float* a, float* b, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = sin_cw(x[i]); b[i] = sin_hw1(x[i]); c[i] = sin_hw2(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), ha(n), hb(n), hc(n);
    for (int i = 0; i < n; ++i) hx[i] = -300.f + 600.f * (float)i / n + 1e-3f * (float)(i % 7);
    float *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, n);
    hipMemcpy(ha.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ec = 0, ea60 = 0, eb60 = 0, ec60 = 0;
    for (int i = 0; i < n; ++i) {
        const double r = sin((double)hx[i]);
        const double da_ = fabs(ha[i] - r), db_ = fabs(hb[i] - r), dc_ = fabs(hc[i] - r);
        ea = fmax(ea, da_); eb = fmax(eb, db_); ec = fmax(ec, dc_);
        if (fabs(hx[i]) < 60) { ea60 = fmax(ea60, da_); eb60 = fmax(eb60, db_); ec60 = fmax(ec60, dc_); }
    }
    printf("max abs err |x|<300 : cody-waite+poly %.3e   v_sin(x/2pi) %.3e   reduce+v_sin %.3e\n", ea, eb, ec);
    printf("max abs err |x|<60  : cody-waite+poly %.3e   v_sin(x/2pi) %.3e   reduce+v_sin %.3e\n", ea60, eb60, ec60);
    return 0;
}
