// P2 upfirdn2d forward (pad -> zero-upsample -> FIR -> decimate) for gfx950.
// Replaces _plugin.upfirdn2d (lib/components/ops/upfirdn2d.cpp:16); per-element definition follows the gather
// form of lib/components/ops/upfirdn2d.cu:29-92: for output (ox, oy) only the taps that land on a real input
// sample of the zero-stuffed signal are visited, so the cost is ceil(fw/upx)*ceil(fh/upy) MACs per output.
// Strides are explicit (NCHW or channels_last).  The filter (<= 1 K taps) is staged in LDS once per workgroup.
//
// Three kernels:
//   upfirdn2d_poly     the StyleGAN resampling filters on dense NCHW planes: 4-tap filters (4x4, or the 4x1 / 1x4 passes of a
//                      separable one) with 2x up, 2x down or neither.  Every resampling constant is a template parameter, so the
//                      polyphase structure is resolved at compile time: a thread owns a 4 x 4 output patch, pulls the
//                      (at most 4 x 4 for 2x up) input samples it touches from the LDS tile into registers once and runs only
//                      the taps that land on a real sample (2 x 2 of the 4 x 4 for 2x up), no integer division anywhere,
//                      filter coefficients in scalar registers, 16-byte stores.
//   upfirdn2d_tiled    dense NCHW planes (the usual case): a workgroup owns a 64 x 16 output tile of one (b, c) plane, stages
//                      the input samples the tile touches in LDS once (zero-filled outside the image: the padding) and every
//                      thread gathers its taps from there -- each input sample is read from HBM once per tile instead of
//                      once per tap, with 32-bit index arithmetic only.  HBM-bound: bytes = input + output once.
//   upfirdn2d_kernel   any strides / very large resampling footprints: one thread per output element, global gathers.
#include "common.hpp"
#include <hip/hip_fp16.h>
#include <type_traits>

namespace {

struct Params {
    int B, C, H, W, fh, fw, outH, outW, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
    int64_t xs[4], ys[4];
};

__device__ __forceinline__ int floor_div(int a, int b) {
    const int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

template <typename T, typename A>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                        T* __restrict__ y, Params p) {
    extern __shared__ float sf[];
    for (int i = threadIdx.x; i < p.fh * p.fw; i += blockDim.x) {
        // store in "correlation order": sf[ky][kx] multiplies padded sample (oy*down + ky, ox*down + kx)
        const int ky = i / p.fw, kx = i % p.fw;
        const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
        sf[i] = f[sy * p.fw + sx];
    }
    __syncthreads();
    const int64_t total = (int64_t)p.B * p.C * p.outH * p.outW;
    // iterate with ox fastest for NCHW-contiguous outputs; general strides still handled
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.outW);
        const int oy = (int)((i / p.outW) % p.outH);
        const int c = (int)((i / ((int64_t)p.outW * p.outH)) % p.C);
        const int b = (int)(i / ((int64_t)p.outW * p.outH * p.C));
        // padded/upsampled coordinate u = o*down + k must satisfy u - pad0 = in*up
        const int bx = ox * p.downx - p.padx0, by = oy * p.downy - p.pady0;
        // smallest kx >= 0 with (bx + kx) % upx == 0
        int kx0 = ((-bx) % p.upx + p.upx) % p.upx;
        int ky0 = ((-by) % p.upy + p.upy) % p.upy;
        const T* __restrict__ xb = x + b * p.xs[0] + c * p.xs[1];
        A acc = 0;
        for (int ky = ky0; ky < p.fh; ky += p.upy) {
            const int iy = floor_div(by + ky, p.upy);
            if (iy < 0 || iy >= p.H) continue;
            for (int kx = kx0; kx < p.fw; kx += p.upx) {
                const int ix = floor_div(bx + kx, p.upx);
                if (ix < 0 || ix >= p.W) continue;
                acc += (A)xb[iy * p.xs[2] + ix * p.xs[3]] * (A)sf[ky * p.fw + kx];
            }
        }
        y[b * p.ys[0] + c * p.ys[1] + oy * p.ys[2] + ox * p.ys[3]] = (T)(acc * (A)p.gain);     // gain last, as upfirdn2d.cu
    }
}

constexpr int64_t kStreamBytes = int64_t(64) << 20;      // outputs from this size on are written with non-temporal stores
constexpr int kTileW = 64, kTileH = 16;          // output tile of the tiled kernel (256 threads x 4 rows each)
constexpr size_t kMaxTileLds = 60 * 1024;        // staged input tile + filter must fit the default dynamic-LDS limit

struct TileGeom { int in_w, in_h; };

// input footprint of a kTileW x kTileH output tile (independent of the tile's position up to +1)
__host__ __device__ inline TileGeom tile_geom(const Params& p) {
    TileGeom g;
    g.in_w = ((kTileW - 1) * p.downx + p.fw - 1) / p.upx + 2;
    g.in_h = ((kTileH - 1) * p.downy + p.fh - 1) / p.upy + 2;
    return g;
}

template <typename T, typename A>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                       Params p, TileGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    A* tile = reinterpret_cast<A*>(smem);                                         // [in_h][in_w]
    float* sf = reinterpret_cast<float*>(smem + sizeof(A) * g.in_w * g.in_h);      // [fh][fw] in correlation order
    const int t = threadIdx.x;
    for (int i = t; i < p.fh * p.fw; i += 256) {
        const int ky = i / p.fw, kx = i % p.fw;
        const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
        sf[i] = f[sy * p.fw + sx];
    }
    const int ox0 = blockIdx.x * kTileW, oy0 = blockIdx.y * kTileH;
    const int plane = blockIdx.z;                                                  // b * C + c, dense NCHW
    // first input sample any tap of this tile can touch
    const int ix0 = floor_div(ox0 * p.downx - p.padx0, p.upx), iy0 = floor_div(oy0 * p.downy - p.pady0, p.upy);
    const T* __restrict__ xb = x + (int64_t)plane * p.H * p.W;
    for (int i = t; i < g.in_w * g.in_h; i += 256) {
        const int ly = i / g.in_w, lx = i - ly * g.in_w;
        const int iy = iy0 + ly, ix = ix0 + lx;
        tile[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (A)xb[iy * p.W + ix] : (A)0;
    }
    __syncthreads();
    const int tx = t & (kTileW - 1), ty0 = t / kTileW;                             // 4 thread rows, 4 output rows each
    const int ox = ox0 + tx;
    if (ox >= p.outW) return;
    const int bx = ox * p.downx - p.padx0;
    const int kx0 = ((-bx) % p.upx + p.upx) % p.upx;                               // first tap that lands on a real sample
    // taps kx0, kx0 + upx, .. read CONSECUTIVE input samples starting at (bx + kx0) / upx (an exact division): the only
    // integer divisions of a thread are these two per axis, none inside the tap loops
    const int lx0 = (bx + kx0) / p.upx - ix0;
    const A gain = (A)p.gain;
    T* __restrict__ yb = y + (int64_t)plane * p.outH * p.outW;
#pragma unroll
    for (int r = 0; r < kTileH / 4; ++r) {
        const int oy = oy0 + ty0 + 4 * r;
        if (oy >= p.outH) break;
        const int by = oy * p.downy - p.pady0;
        const int ky0 = ((-by) % p.upy + p.upy) % p.upy;
        const A* row = tile + ((by + ky0) / p.upy - iy0) * g.in_w + lx0;
        A acc = 0;
        for (int ky = ky0; ky < p.fh; ky += p.upy, row += g.in_w) {
            const float* fr = sf + ky * p.fw + kx0;
            const A* px = row;
            for (int kx = kx0; kx < p.fw; kx += p.upx, fr += p.upx, ++px) acc += *px * (A)*fr;
        }
        yb[oy * p.outW + ox] = (T)(acc * gain);                                     // gain last, as upfirdn2d.cu
    }
}

// ---- polyphase kernel: compile-time resampling constants -------------------------------------------------------------------
// Output o reads padded / zero-stuffed coordinate u = o*D - pad0 + k for tap k (correlation order), which is the real input
// sample (u / U) when U divides u.  A thread's patch starts at an output index that is a multiple of V with V*D % U == 0, so
// u = (q*U + R) + v*D + k with the launch-uniform phase R = (-pad0) mod U: tap k of patch output v is live iff
// (R + v*D + k) % U == 0 and reads register (R + v*D + k) / U of the thread's window, all compile-time.
template <int UX_, int UY_, int DX_, int DY_, int FW_, int FH_, int RX_, int RY_, int VY_ = 4>
struct Poly {
    static constexpr int UX = UX_, UY = UY_, DX = DX_, DY = DY_, FW = FW_, FH = FH_, RX = RX_, RY = RY_;
#ifndef H3D_UPFIRDN_TW
#define H3D_UPFIRDN_TW 64
#endif
    // patch per thread (4 rows: 0.208 -> 0.190 ms vs 2), tile per WG: TW outputs wide (TW / 4 threads across), 256 / (TW / 4) thread rows
    static constexpr int VX = 4, VY = VY_, TW = H3D_UPFIRDN_TW, TXG = TW / VX, TH = (256 / TXG) * VY_;
    static_assert(TW % VX == 0 && 256 % TXG == 0, "tile width");
    static constexpr int WX = (RX + (VX - 1) * DX + FW - 1) / UX + 1;                 // register window of a thread
    static constexpr int WY = (RY + (VY - 1) * DY + FH - 1) / UY + 1;
    static constexpr int IN_W = (RX + (TW - 1) * DX + FW - 1) / UX + 1;               // LDS tile of a workgroup
    static constexpr int IN_H = (RY + (TH - 1) * DY + FH - 1) / UY + 1;
    static constexpr int LD = (IN_W + 3) / 4 * 4 + 4;                                  // row stride: multiple of 4, rows 4 banks apart
    static_assert((VX * DX) % UX == 0 && (VY * DY) % UY == 0, "patch origin must keep the phase");
};

template <typename T, typename P, bool NT>
__global__ __launch_bounds__(256) void upfirdn2d_poly(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                      Params p, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float tile[P::IN_H * P::LD];
    float c[P::FH][P::FW];                                                           // correlation order; uniform -> scalar registers
#pragma unroll
    for (int ky = 0; ky < P::FH; ++ky)
#pragma unroll
        for (int kx = 0; kx < P::FW; ++kx)
            c[ky][kx] = f[(p.flip ? ky : P::FH - 1 - ky) * P::FW + (p.flip ? kx : P::FW - 1 - kx)];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ox0 = blockIdx.x * P::TW, oy0 = blockIdx.y * P::TH;
    const int ix0 = floor_div(ox0 * P::DX - p.padx0, P::UX), iy0 = floor_div(oy0 * P::DY - p.pady0, P::UY);
    const T* __restrict__ xb = x + (int64_t)blockIdx.z * p.H * p.W;
    for (int ly = wave; ly < P::IN_H; ly += 4) {                                     // a wave stages whole rows: no index division
        const int iy = iy0 + ly;
        const bool row_ok = iy >= 0 && iy < p.H;
        const T* __restrict__ xr = xb + (int64_t)iy * p.W;
        for (int lx = lane; lx < P::IN_W; lx += 64) {
            const int ix = ix0 + lx;
            tile[ly * P::LD + lx] = (row_ok && ix >= 0 && ix < p.W) ? (float)xr[ix] : 0.f;
        }
    }
    __syncthreads();
    const int txg = t % P::TXG, ty = t / P::TXG;
    const int ox = ox0 + P::VX * txg, oy = oy0 + P::VY * ty;
    if (ox >= p.outW || oy >= p.outH) return;
    const float* __restrict__ wp = tile + (P::VY * ty * P::DY / P::UY) * P::LD + P::VX * txg * P::DX / P::UX;
    float win[P::WY][P::WX];
#pragma unroll
    for (int j = 0; j < P::WY; ++j)
#pragma unroll
        for (int i = 0; i < P::WX; ++i) win[j][i] = wp[j * P::LD + i];
    T* __restrict__ yb = y + (int64_t)blockIdx.z * p.outH * p.outW;
#pragma unroll
    for (int vy = 0; vy < P::VY; ++vy) {
        if (oy + vy >= p.outH) break;
        float acc[P::VX];
#pragma unroll
        for (int vx = 0; vx < P::VX; ++vx) {
            float a = 0.f;                                                           // ky outer, kx inner: the order of the other kernels
#pragma unroll
            for (int ky = 0; ky < P::FH; ++ky) {
                if ((P::RY + vy * P::DY + ky) % P::UY) continue;
#pragma unroll
                for (int kx = 0; kx < P::FW; ++kx) {
                    if ((P::RX + vx * P::DX + kx) % P::UX) continue;
                    a += win[(P::RY + vy * P::DY + ky) / P::UY][(P::RX + vx * P::DX + kx) / P::UX] * c[ky][kx];
                }
            }
            acc[vx] = a * p.gain;                                                    // gain last, as upfirdn2d.cu
        }
        T* o = yb + (int64_t)(oy + vy) * p.outW + ox;
        if (vec_ok && ox + P::VX <= p.outW) {
            typedef T vecT __attribute__((ext_vector_type(4)));
            // NT: streaming (non-temporal) stores for an output larger than the caches (round 6: 3.5 -> 5.2 TB/s on the 2x-up shape
            // of the roofline table, profiles/r6_ops_nt_stores.txt).  A template parameter, not a run-time branch: the optimiser
            // merges `if (c) nt-store else store` into ONE plain store.
            if constexpr (NT) __builtin_nontemporal_store(vecT{(T)acc[0], (T)acc[1], (T)acc[2], (T)acc[3]}, reinterpret_cast<vecT*>(o));
            else *reinterpret_cast<vecT*>(o) = vecT{(T)acc[0], (T)acc[1], (T)acc[2], (T)acc[3]};
        } else {
#pragma unroll
            for (int vx = 0; vx < P::VX; ++vx)
                if (ox + vx < p.outW) o[vx] = (T)acc[vx];
        }
    }
}

inline int pos_mod(int a, int m) { return ((a % m) + m) % m; }

template <typename T, typename P>
int launch_poly(const void* x, const float* f, void* y, const Params& p, hipStream_t st) {
    const int64_t planes = (int64_t)p.B * p.C;
    const int vec_ok = (p.outW % 4 == 0) && (reinterpret_cast<uintptr_t>(y) % (4 * sizeof(T)) == 0);
    const bool stream = vec_ok && planes * p.outH * p.outW * (int64_t)sizeof(T) >= kStreamBytes;      // streaming stores (see the kernel)
    for (int64_t z0 = 0; z0 < planes; z0 += 65535) {
        const unsigned nz = (unsigned)((planes - z0) < 65535 ? (planes - z0) : 65535);
        const dim3 grid((p.outW + P::TW - 1) / P::TW, (p.outH + P::TH - 1) / P::TH, nz);
        h3d::pre_launch();
        if (stream) hipLaunchKernelGGL((upfirdn2d_poly<T, P, true>), grid, dim3(256), 0, st, (const T*)x + z0 * p.H * p.W, f,
                                       (T*)y + z0 * p.outH * p.outW, p, vec_ok);
        else hipLaunchKernelGGL((upfirdn2d_poly<T, P, false>), grid, dim3(256), 0, st, (const T*)x + z0 * p.H * p.W, f,
                                (T*)y + z0 * p.outH * p.outW, p, vec_ok);
        const int rc = h3d::launch_status("h3d_upfirdn2d");
        if (rc) return rc;
    }
    return H3D_OK;
}

// one resampling geometry, every phase of it
template <typename T, int UX, int UY, int DX, int DY, int FW, int FH, int VY>
int poly_phases(const void* x, const float* f, void* y, const Params& p, hipStream_t st) {
    const int rx = pos_mod(-p.padx0, UX), ry = pos_mod(-p.pady0, UY);
    if constexpr (UX == 2 && UY == 2) {
        if (rx == 0 && ry == 0) return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 0, VY>>(x, f, y, p, st);
        if (rx == 1 && ry == 0) return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 1, 0, VY>>(x, f, y, p, st);
        if (rx == 0 && ry == 1) return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 1, VY>>(x, f, y, p, st);
        return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 1, 1, VY>>(x, f, y, p, st);
    } else if constexpr (UX == 2) {
        if (rx == 0) return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 0, VY>>(x, f, y, p, st);
        return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 1, 0, VY>>(x, f, y, p, st);
    } else if constexpr (UY == 2) {
        if (ry == 0) return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 0, VY>>(x, f, y, p, st);
        return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 1, VY>>(x, f, y, p, st);
    } else {
        return launch_poly<T, Poly<UX, UY, DX, DY, FW, FH, 0, 0, VY>>(x, f, y, p, st);
    }
}

// -> H3D_OK / error when the geometry is one of the compiled ones (handled), -1 when it is not
template <typename T>
int try_poly(const void* x, const float* f, void* y, const Params& p, hipStream_t st) {
    if ((int64_t)p.H * p.W >= (1ll << 31) || (int64_t)p.outH * p.outW >= (1ll << 31)) return -1;
#define H3D_POLY(UX, UY, DX, DY, FW, FH)                                                                              \
    if (p.upx == UX && p.upy == UY && p.downx == DX && p.downy == DY && p.fw == FW && p.fh == FH)                    \
        return poly_phases<T, UX, UY, DX, DY, FW, FH, (DY == 2 ? 2 : 4)>(x, f, y, p, st);      /* 2x down: a 64-row tile's input would not fit 64 KB */
    H3D_POLY(2, 2, 1, 1, 4, 4) H3D_POLY(1, 1, 2, 2, 4, 4) H3D_POLY(1, 1, 1, 1, 4, 4)      // 2-D filter
    H3D_POLY(2, 1, 1, 1, 4, 1) H3D_POLY(1, 1, 2, 1, 4, 1) H3D_POLY(1, 1, 1, 1, 4, 1)      // separable: the row pass
    H3D_POLY(1, 2, 1, 1, 1, 4) H3D_POLY(1, 1, 1, 2, 1, 4) H3D_POLY(1, 1, 1, 1, 1, 4)      //            the column pass
#undef H3D_POLY
    return -1;
}

bool dense_nchw(const Params& p) {
    return p.xs[3] == 1 && p.xs[2] == p.W && p.xs[1] == (int64_t)p.H * p.W && p.xs[0] == (int64_t)p.C * p.H * p.W &&
           p.ys[3] == 1 && p.ys[2] == p.outW && p.ys[1] == (int64_t)p.outH * p.outW && p.ys[0] == (int64_t)p.C * p.outH * p.outW;
}

template <typename T, typename A>
int launch(const void* x, const float* f, void* y, const Params& p, hipStream_t st) {
    if constexpr (!std::is_same<T, double>::value) {
        if (dense_nchw(p)) {
            const int rc = try_poly<T>(x, f, y, p, st);
            if (rc >= 0) return rc;
        }
    }
    const TileGeom g = tile_geom(p);
    const int64_t planes = (int64_t)p.B * p.C;
    const size_t lds = sizeof(A) * g.in_w * g.in_h + sizeof(float) * p.fh * p.fw;
    if (dense_nchw(p) && lds <= kMaxTileLds && (int64_t)p.H * p.W < (1ll << 31) && (int64_t)p.outH * p.outW < (1ll << 31)) {
        // blockIdx.z is limited to 65535: fold the planes over several launches if ever needed
        for (int64_t z0 = 0; z0 < planes; z0 += 65535) {
            const unsigned nz = (unsigned)((planes - z0) < 65535 ? (planes - z0) : 65535);
            h3d::pre_launch();
            hipLaunchKernelGGL((upfirdn2d_tiled<T, A>), dim3((p.outW + kTileW - 1) / kTileW, (p.outH + kTileH - 1) / kTileH, nz),
                               dim3(256), lds, st, (const T*)x + z0 * p.H * p.W, f, (T*)y + z0 * p.outH * p.outW, p, g);
            const int rc = h3d::launch_status("h3d_upfirdn2d");
            if (rc) return rc;
        }
        return H3D_OK;
    }
    const int64_t total = (int64_t)p.B * p.C * p.outH * p.outW;
    const int64_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? (want < 1 ? 1 : want) : 256 * 32);
    h3d::pre_launch();
    hipLaunchKernelGGL((upfirdn2d_kernel<T, A>), dim3(grid), dim3(256), sizeof(float) * p.fh * p.fw, st, (const T*)x, f,
                       (T*)y, p);
    return h3d::launch_status("h3d_upfirdn2d");
}

}  // namespace

extern "C" int h3d_upfirdn2d(const void* x, const float* f, void* y, int dtype, int B, int C, int H, int W,
                             const int64_t xs[4], int fh, int fw, int outH, int outW, const int64_t ys[4], int upx,
                             int upy, int downx, int downy, int padx0, int pady0, int flip, float gain,
                             h3d_stream_t stream) {
    H3D_REQUIRE(x && f && y && xs && ys, "h3d_upfirdn2d: null pointer");
    H3D_REQUIRE(dtype >= 0 && dtype <= 2, "h3d_upfirdn2d: dtype %d (0=f32,1=f16,2=f64)", dtype);
    H3D_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1, "h3d_upfirdn2d: x has zero size");
    H3D_REQUIRE(fh >= 1 && fw >= 1, "h3d_upfirdn2d: f must be at least 1x1");
    H3D_REQUIRE(fh * fw <= 8192, "h3d_upfirdn2d: filter too large");
    H3D_REQUIRE(upx >= 1 && upy >= 1, "h3d_upfirdn2d: upsampling factor must be at least 1");
    H3D_REQUIRE(downx >= 1 && downy >= 1, "h3d_upfirdn2d: downsampling factor must be at least 1");
    H3D_REQUIRE(outH >= 1 && outW >= 1, "h3d_upfirdn2d: output must be at least 1x1");
    Params p;
    p.B = B; p.C = C; p.H = H; p.W = W; p.fh = fh; p.fw = fw; p.outH = outH; p.outW = outW;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0; p.flip = flip;
    p.gain = gain;
    for (int i = 0; i < 4; ++i) { p.xs[i] = xs[i]; p.ys[i] = ys[i]; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 0) return launch<float, float>(x, f, y, p, st);
    if (dtype == 1) return launch<_Float16, float>(x, f, y, p, st);
    return launch<double, double>(x, f, y, p, st);
}
