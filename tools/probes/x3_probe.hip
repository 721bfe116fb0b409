// Stand-alone probe of the MFMA f16 32x32x16 operand/accumulator layout and of the permlane32_swap re-layout
// used by csrc/field_x3.hip.  Build: hipcc --offload-arch=gfx950 -O2 x3_probe.hip -o x3_probe ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// D[i][j] = sum_k A[i][k] B[k][j], A 32x16 (row-major), B 16x32 (row-major); assumed slot map k = 8*(l>>5)+e
__global__ void probe_gemm(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 16 + 8 * h + e]; b[e] = (_Float16)B[(8 * h + e) * 32 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r >> 2) * 8 + 4 * h + (r & 3);
        D[row * 32 + i] = acc[r];
    }
}

// Take a 32x32 matrix M[n][m] in accumulator layout (lane: col m, rows by reg), re-lay it out as two B fragments
// (k-steps 0 and 1, k = n) with the swap trick, and write back what each lane believes it holds: F[ks][lane][e]
__global__ void probe_swap(const float* M, float* F) {
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    unsigned H[4][2];
    for (int rg = 0; rg < 4; ++rg) {
        _Float16 v[4];
        for (int q = 0; q < 4; ++q) v[q] = (_Float16)M[(rg * 8 + 4 * h + q) * 32 + m];
        H[rg][0] = __builtin_bit_cast(unsigned, half2v{v[0], v[1]});
        H[rg][1] = __builtin_bit_cast(unsigned, half2v{v[2], v[3]});
    }
    for (int pr = 0; pr < 2; ++pr) {
        u32x4 fh;
        for (int c = 0; c < 2; ++c) {
            auto a = __builtin_amdgcn_permlane32_swap(H[2 * pr][c], H[2 * pr + 1][c], false, false);
            fh[c] = a[0]; fh[2 + c] = a[1];
        }
        half8 f = __builtin_bit_cast(half8, fh);
        for (int e = 0; e < 8; ++e) F[(pr * 64 + l) * 8 + e] = (float)f[e];
    }
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32], hM[32 * 32], hF[2 * 64 * 8];
    for (int i = 0; i < 32 * 16; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j];
        ref[i * 32 + j] = s;
        hM[i * 32 + j] = (float)(i * 32 + j);      // value encodes (n, m): n = v / 32, m = v % 32 (exact in f16 up to 2048)
    }
    float *dA, *dB, *dD, *dM, *dF;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dM, sizeof hM); hipMalloc(&dF, sizeof hF);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dM, hM, sizeof hM, hipMemcpyHostToDevice);
    probe_gemm<<<1, 64>>>(dA, dB, dD);
    probe_swap<<<1, 64>>>(dM, dF);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hF, dF, sizeof hF, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) if (fabsf(hD[i] - ref[i]) > 1e-3f) ++bad;
    printf("gemm: %d / 1024 mismatches (slot map k=8h+e, D map row=8(r>>2)+4h+(r&3))\n", bad);
    if (bad) { for (int i = 0; i < 4; ++i) { for (int j = 0; j < 8; ++j) printf("%7.1f/%7.1f ", hD[i*32+j], ref[i*32+j]); printf("\n"); } }
    // expected fragment: lane (m, h), k-step pr, element e  -> n = 16*pr + 8*h + e, same m
    int badf = 0;
    for (int pr = 0; pr < 2; ++pr) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
        const int m = l & 31, h = l >> 5, n = 16 * pr + 8 * h + e;
        if (hF[(pr * 64 + l) * 8 + e] != (float)(n * 32 + m)) ++badf;
    }
    printf("swap re-layout: %d / 1024 mismatches\n", badf);
    if (badf) for (int l : {0, 1, 32, 33}) { printf("lane %d ks0:", l); for (int e = 0; e < 8; ++e) { float v = hF[l * 8 + e]; printf(" (n=%d,m=%d)", (int)v / 32, (int)v % 32); } printf("\n"); }
    return 0;
}
