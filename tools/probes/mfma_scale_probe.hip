// Stand-alone probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: operand slot pairing, scale semantics, fp8 (e4m3) and
// fp6 (e2m3) element packing, and the fp32 -> fp8 / fp6 conversion instructions the x2 engines use.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_scale_probe.hip -o mfma_scale_probe ; run on a GPU box.
//
// Hypothesis H (what the engines rely on): with lane l = 32*h + i holding 32 element slots s = 0..31 of operand A (row i)
// and lane l = 32*h + j holding 32 slots of operand B (column j),
//     D[i][j] = 2^(sa-127) * 2^(sb-127) * sum_{h in 0,1} sum_{s<32} A_lane(32h+i)[s] * B_lane(32h+j)[s]
// i.e. slot s of lane-half h of A contracts with slot s of lane-half h of B (the absolute k index is irrelevant to us),
// D in the standard 32x32 accumulator layout row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FMT>
__global__ void probe(const int* A, const int* B, float* D, int sa, int sb) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[l * 8 + e]; b[e] = B[l * 8 + e]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT, FMT, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = acc[r];
}

// pairing matrix: problem (pa, pb) = blockIdx: A one-hot (value 1.0) at lane 32*(pa>>5), slot pa&31; B one-hot at lane
// 32*(pb>>5), slot pb&31; P[pa][pb] = D[0][0]
__global__ void pairing(float* P) {
    const int l = threadIdx.x, pa = blockIdx.x, pb = blockIdx.y;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
        if (l == 32 * (pa >> 5) && e == (pa & 31) / 4) a[e] = 0x38 << (8 * (pa & 3));
        if (l == 32 * (pb >> 5) && e == (pb & 31) / 4) b[e] = 0x38 << (8 * (pb & 3));
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    if (l == 0) P[pa * 64 + pb] = acc[0];
}

// conversions: fp32 pair -> packed fp8 (e4m3) with a scale; results written raw
__global__ void cvt_probe(const float* x, int* out, float scale) {
    const int l = threadIdx.x;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 v = {0, 0};
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, x[2 * l], x[2 * l + 1], scale, false);
    out[l] = __builtin_bit_cast(int, v);
}

// issue-rate / power micro-benchmark: N dependent-free MFMAs on 4 accumulators, one wave per SIMD, random operands
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void rate(const int* A, float* out, int n) {
    const int l = threadIdx.x & 63;
    i32x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[l * 8 + e]; b[e] = A[(63 - l) * 8 + e]; }
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 0) {          // f16 32x32x16
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const half8 ha = __builtin_bit_cast(half8, i32x4{a[0], a[1], a[2], a[3]});
                const half8 hb = __builtin_bit_cast(half8, i32x4{b[0], b[1], b[2], b[3]});
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[q], 0, 0, 0);
            } else if (MODE == 1) {   // scaled fp8 32x32x64
                acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
            } else {                  // scaled fp6 32x32x64
                acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 2, 2, 0, 0x74747474, 0, 0x7f7f7f7f);
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
    return s ? -r : r;
}
static float e2m3(uint8_t v) {          // 6 bits: s eem mmm, bias 1
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    float r = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return s ? -r : r;
}

int main() {
    int *dA, *dB; float *dD, *dP;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dD, 64 * 16 * 4); hipMalloc(&dP, 64 * 64 * 4);
    srand(1);
    int bad_total = 0;
    for (int fmt = 0; fmt <= 2; fmt += 2) {                  // 0 = fp8 e4m3, 2 = fp6 e2m3
        uint8_t hA[64][32], hB[64][32];                      // element codes per lane / slot
        int rA[64][8], rB[64][8];
        memset(rA, 0, sizeof rA); memset(rB, 0, sizeof rB);
        for (int l = 0; l < 64; ++l)
            for (int s = 0; s < 32; ++s) {
                if (fmt == 0) {
                    do { hA[l][s] = rand() & 255; } while ((hA[l][s] & 0x7f) == 0x7f || (hA[l][s] & 0x78) > 0x48);
                    do { hB[l][s] = rand() & 255; } while ((hB[l][s] & 0x7f) == 0x7f || (hB[l][s] & 0x78) > 0x48);
                    rA[l][s / 4] |= hA[l][s] << (8 * (s & 3));
                    rB[l][s / 4] |= hB[l][s] << (8 * (s & 3));
                } else {
                    hA[l][s] = rand() & 63; hB[l][s] = rand() & 63;
                    const int bit = 6 * s;                      // little-endian bit stream over 6 dwords
                    uint64_t va = (uint64_t)hA[l][s] << (bit & 31), vb = (uint64_t)hB[l][s] << (bit & 31);
                    rA[l][bit / 32] |= (int)(uint32_t)va; rB[l][bit / 32] |= (int)(uint32_t)vb;
                    if ((bit & 31) > 26) { rA[l][bit / 32 + 1] |= (int)(va >> 32); rB[l][bit / 32 + 1] |= (int)(vb >> 32); }
                }
            }
        hipMemcpy(dA, rA, sizeof rA, hipMemcpyHostToDevice);
        hipMemcpy(dB, rB, sizeof rB, hipMemcpyHostToDevice);
        const int sas[3] = {127, 116, 130}, sbs[3] = {127, 127, 120};
        for (int t = 0; t < 3; ++t) {
            const int sa = sas[t] * 0x01010101, sb = sbs[t] * 0x01010101;
            if (fmt == 0) probe<0><<<1, 64>>>(dA, dB, dD, sa, sb); else probe<2><<<1, 64>>>(dA, dB, dD, sa, sb);
            float hD[64][16];
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            int bad = 0; double worst = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                    double acc = 0;
                    for (int h = 0; h < 2; ++h)
                        for (int s = 0; s < 32; ++s)
                            acc += (double)(fmt == 0 ? e4m3(hA[32 * h + i][s]) : e2m3(hA[32 * h + i][s])) *
                                   (double)(fmt == 0 ? e4m3(hB[32 * h + j][s]) : e2m3(hB[32 * h + j][s]));
                    acc *= ldexp(1.0, sas[t] - 127 + sbs[t] - 127);
                    const double err = fabs(acc - hD[l][r]) / (fabs(acc) + 1e-3);
                    if (err > 1e-5) ++bad;
                    if (err > worst) worst = err;
                }
            printf("fmt %d (%s) scale_a 2^%d scale_b 2^%d: hypothesis H %s  (mismatches %d / 1024, worst rel %.2e)\n", fmt,
                   fmt == 0 ? "fp8 e4m3" : "fp6 e2m3", sas[t] - 127, sbs[t] - 127, bad ? "FAILS" : "holds", bad, worst);
            bad_total += bad;
        }
    }
    // slot pairing by one-hot operands (fp8)
    pairing<<<dim3(64, 64), 64>>>(dP);
    static float P[64][64];
    hipMemcpy(P, dP, sizeof P, hipMemcpyDeviceToHost);
    int ident = 1;
    for (int a = 0; a < 64; ++a) for (int b = 0; b < 64; ++b) if ((P[a][b] != 0.f) != (a == b)) ident = 0;
    printf("one-hot slot pairing: %s\n", ident ? "identity (slot s, half h of A <-> slot s, half h of B)" : "NOT identity:");
    if (!ident)
        for (int a = 0; a < 64; ++a) { printf("  A slot %2d.%2d <-> B:", a >> 5, a & 31); for (int b = 0; b < 64; ++b) if (P[a][b] != 0.f) printf(" %d.%d(%g)", b >> 5, b & 31, P[a][b]); printf("\n"); }
    // conversion probe
    float hx[128]; int hout[64]; float* dx; int* dout;
    for (int i = 0; i < 128; ++i) hx[i] = ((rand() & 0xffff) / 65536.f - 0.5f) * ldexpf(1.f, (rand() % 12) - 8);
    hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof hout);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    for (int t = 0; t < 2; ++t) {
        const float scale = t ? ldexpf(1.f, -4) : 1.f;
        cvt_probe<<<1, 64>>>(dx, dout, scale);
        hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
        // which of x*scale / x/scale does the instruction encode?
        double e_mul = 0, e_div = 0;
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 2; ++q) {
                const float got = e4m3((hout[l] >> (8 * q)) & 255), x = hx[2 * l + q];
                e_mul = fmax(e_mul, fabs(got - x * scale) / (fabs(x * scale) + ldexp(1.0, -9)));
                e_div = fmax(e_div, fabs(got - x / scale) / (fabs(x / scale) + ldexp(1.0, -9)));
            }
        printf("cvt_scalef32_pk_fp8_f32 scale %g: low 16 bits = fp8(x0), fp8(x1) of x*scale: worst rel %.3f ; of x/scale: worst rel %.3f\n",
               scale, e_mul, e_div);
        if (t) printf("   sample: x=%g,%g -> bytes %02x %02x (%g, %g) upper half %04x\n", hx[0], hx[1], hout[0] & 255, (hout[0] >> 8) & 255,
                      e4m3(hout[0] & 255), e4m3((hout[0] >> 8) & 255), (hout[0] >> 16) & 0xffff);
    }
    // rates: 1024 workgroups of 4 waves (4 per CU), n iterations x 4 MFMAs per wave
    {
        int rnd[64 * 8];
        for (int i = 0; i < 64 * 8; ++i) {          // bytes 0x28..0x47 / 0xa8..: normal e4m3 values of moderate size, also sane f16 halves
            unsigned w = 0;
            for (int q = 0; q < 4; ++q) w |= (unsigned)(0x28 + (rand() % 32) + ((rand() & 1) << 7)) << (8 * q);
            rnd[i] = (int)w;
        }
        hipMemcpy(dA, rnd, sizeof rnd, hipMemcpyHostToDevice);
        float* dout2; hipMalloc(&dout2, 2048 * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[3] = {"v_mfma_f32_32x32x16_f16", "v_mfma_scale_f32_32x32x64_f8f6f4 fp8", "v_mfma_scale_f32_32x32x64_f8f6f4 fp6"};
        const double flops[3] = {2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 64, 2.0 * 32 * 32 * 64};
        for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 3; ++m) {
            const int n = m == 0 ? 12000000 : 6000000, grid = 1024;
            hipEventRecord(e0);
            if (m == 0) rate<0><<<grid, 256>>>(dA, dout2, n); else if (m == 1) rate<1><<<grid, 256>>>(dA, dout2, n); else rate<2><<<grid, 256>>>(dA, dout2, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = (double)grid * 4 * n * 4;
            printf("rate %-42s %8.1f ms  %7.1f TFLOP/s  %6.2f ns per MFMA per SIMD (x 2.4 GHz = %5.1f cyc at full clock)\n", names[m], ms,
                   nm * flops[m] / ms / 1e9, ms * 1e6 / (nm / 1024.0), ms * 1e6 / (nm / 1024.0) * 2.4);
            fflush(stdout);
        }
    }
    return bad_total ? 1 : 0;
}
