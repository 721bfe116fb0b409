// A5 pose-conditioned FiLM-SIREN (COORDCONCATSIREN) and the fused A5+A6 render kernel for gfx950.
//
// Reference semantics: lib/implicit_funcitions/modulated.py:41-75, lib/components/pigan_layers.py:63-87,
// lib/generators/volume_rendering.py:12-56.
//
// One 256-thread workgroup walks one *group* of points: 64 points (one tile) for the stand-alone field, or
// max(64, S) points -- whole rays -- for the fused render.  All eight dense layers of a tile run on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, the 1e-3 parity budget does not survive fp16/bf16 inputs
// through the freq~45 sines), activations never leave LDS, weights stream from L2 in MFMA fragment order
// (field_common.hpp).  Density (N=1) and colour (N=3) heads are VALU dot products over the LDS tile.  In the
// fused kernel wave 0 turns the 64 densities into compositing weights with a (segmented) wavefront product scan
// and the feature GEMM's accumulators are reduced over the rows of each ray in registers, so the [N, F+4]
// field tensor (1 KB per sample) never exists in HBM.
//
// MFMA-bound: 2*(7*Hd^2 + 41*Hd) flop per sample.
#include "field_common.hpp"

using namespace h3d;

namespace {

struct Layout {   // offsets in floats into the packed blob
    int HdP, FP, NT, NTF, KBH;
    int64_t w_coord, w_geo, w_film[4], w_color, w_feat;
    int64_t b_coord, b_geo, b_film[4], b_color, w_dir, b_feat, w_sigma, w_rgb, scal, total;
};

Layout make_layout(int Hd, int F) {
    Layout L;
    L.HdP = round_up(Hd, 32);
    L.FP = round_up(F, 32);
    L.NT = L.HdP / 32;
    L.NTF = L.FP / 32;
    L.KBH = L.HdP / 8;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += n; return r; };
    L.w_coord = take((int64_t)L.NT * 2 * 256);            // K = 3 padded to two k-blocks (even count)
    L.w_geo = take((int64_t)L.NT * 4 * 256);
    L.w_film[0] = take((int64_t)L.NT * 2 * L.KBH * 256);
    for (int k = 1; k < 4; ++k) L.w_film[k] = take((int64_t)L.NT * L.KBH * 256);
    L.w_color = take((int64_t)L.NT * (2 + L.KBH) * 256);  // blocks 0-1: direction rows (+ zero pad), 2..: x rows
    L.w_feat = take((int64_t)L.NTF * L.KBH * 256);
    L.b_coord = take(L.HdP);
    L.b_geo = take(L.HdP);
    for (int k = 0; k < 4; ++k) L.b_film[k] = take(L.HdP);
    L.b_color = take(L.HdP);
    L.w_dir = take(3 * (int64_t)L.HdP);
    L.b_feat = take(L.FP);
    L.w_sigma = take(L.HdP);
    L.w_rgb = take(3 * (int64_t)L.HdP);
    L.scal = take(4);
    L.total = o;
    return L;
}

struct Args {
    const float* blob;
    const float* points;
    const float* geo;
    const float* dirs;
    const float* freq;
    const float* phase;
    float* out;                 // stand-alone: [B,N,F+4]
    const float* z_vals;        // fused
    const float* noise;
    float* feats;
    float* depth;
    float* weights;
    int64_t N;
    int Hd, F, geo_stride, S, clamp_mode, last_back, white_back;
    float input_scaler;
    Layout L;
};

__device__ __forceinline__ float density(float x, int clamp_mode) {
    if (clamp_mode == 1) return x > 20.f ? x : log1pf(expf(x));
    return fmaxf(x, 0.f);
}

struct Film { float b, f, p; };

template <int NTW, bool FUSED>
__global__ __launch_bounds__(kFieldThreads) void field_kernel(Args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Layout& L = A.L;
    const int act_rows = L.HdP > 40 ? L.HdP : 40;
    float* actT = smem;                              // [act_rows][MS]; first 40 rows double as the input tile
    float* dirT = actT + act_rows * kMS;             // [16][MS] (rows 3-15 zero)
    float* part = dirT + 16 * kMS;                   // [4 waves][3][64] partial dot products
    float* wgt = part + 4 * 3 * 64;                  // [64] compositing weights
    float* bgl = wgt + 64;                           // [64] background term of the row's ray
    float* rgbv = bgl + 64;                          // [64][3]
    float* sigv = rgbv + 64 * 3;                     // [64]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int64_t N = A.N;
    const int Hd = A.Hd, F = A.F, S = A.S;
    const float* __restrict__ blob = A.blob;
    const float* __restrict__ fr = A.freq + (int64_t)b * 4 * Hd;
    const float* __restrict__ ph = A.phase + (int64_t)b * 4 * Hd;
    const int group_pts = FUSED ? (S > 64 ? S : 64) : 64;
    const int tiles = group_pts / 64;
    const int64_t g0 = (int64_t)blockIdx.x * group_pts;
    const int seglen = FUSED ? (S < 64 ? S : 64) : 64;

    // state carried across the tiles of a multi-tile ray (wave 0 lanes hold identical copies)
    float carryT = 1.f, carryW = 0.f, carryD = 0.f;
    float rayacc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) rayacc[i] = 0.f;
    float rgbacc = 0.f;

    for (int ti = 0; ti < tiles; ++ti) {
        const int64_t n0 = g0 + (int64_t)ti * 64;
        const bool last_tile = ti == tiles - 1;

        // ---- stage the input tile, transposed: rows 0-2 scaled coords, 8-38 geometry features
        for (int idx = t; idx < 64 * 8; idx += kFieldThreads) {
            const int m = idx & 63, c = idx >> 6;
            const int64_t n = n0 + m;
            actT[c * kMS + m] = (c < 3 && n < N) ? A.points[((int64_t)b * N + n) * 3 + c] * A.input_scaler : 0.f;
        }
        for (int idx = t; idx < 64 * 32; idx += kFieldThreads) {
            const int k = idx & 31, m = idx >> 5;
            const int64_t n = n0 + m;
            actT[(8 + k) * kMS + m] = (k < 31 && n < N) ? A.geo[((int64_t)b * N + n) * A.geo_stride + k] : 0.f;
        }
        if (A.dirs) {
            for (int idx = t; idx < 64 * 16; idx += kFieldThreads) {
                const int m = idx & 63, c = idx >> 6;
                const int64_t n = n0 + m;
                dirT[c * kMS + m] = (c < 3 && n < N) ? A.dirs[((int64_t)b * N + n) * 3 + c] : 0.f;
            }
        }
        __syncthreads();

        // ---- first layers (K = 3 and K = 31, zero padded)
        f32x16 accG[2][NTW], acc[2][NTW];
        zero_acc<NTW>(accG);
        zero_acc<NTW>(acc);
        gemm_phase<NTW>(accG, actT + 8 * kMS, reinterpret_cast<const float4*>(blob + L.w_geo), 4, 0, 4, L.NT, wave, lane);
        gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + L.w_coord), 2, 0, 2, L.NT, wave, lane);
        __syncthreads();
        {
            const float* __restrict__ bb = blob + L.b_coord;
            store_act<NTW>(acc, actT, L.NT, Hd, wave, lane,
                           [&](int n) { return Film{bb[n], 0.f, 0.f}; },
                           [](float v, const Film& c) { return sin_accurate(30.f * (v + c.b)); });
        }
        __syncthreads();

        // ---- FiLM layer 0 in two K halves (coord half, then geometry half) sharing one accumulator
        zero_acc<NTW>(acc);
        gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + L.w_film[0]), L.KBH, 0, 2 * L.KBH, L.NT, wave, lane);
        __syncthreads();
        {
            const float* __restrict__ bb = blob + L.b_geo;
            store_act<NTW>(accG, actT, L.NT, Hd, wave, lane,
                           [&](int n) { return Film{bb[n], 0.f, 0.f}; },
                           [](float v, const Film& c) { return sin_accurate(30.f * (v + c.b)); });
        }
        __syncthreads();
        gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + L.w_film[0]), L.KBH, L.KBH, 2 * L.KBH, L.NT, wave, lane);
        __syncthreads();
        for (int l = 0; l < 4; ++l) {
            if (l > 0) {
                zero_acc<NTW>(acc);
                gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + L.w_film[l]), L.KBH, 0, L.KBH, L.NT, wave, lane);
                __syncthreads();
            }
            const float* __restrict__ bb = blob + L.b_film[l];
            const int off = l * Hd;
            store_act<NTW>(acc, actT, L.NT, Hd, wave, lane,
                           [&](int n) { return Film{bb[n], fr[off + n] * 15.f + 30.f, ph[off + n]}; },
                           [](float v, const Film& c) { return sin_accurate(fmaf(c.f, v + c.b, c.p)); });
            __syncthreads();
        }

        // ---- density head: VALU dot product, wave w covers a quarter of K, lane = point
        {
            const float* __restrict__ ws = blob + L.w_sigma;
            const int kq = L.HdP / 4, k0 = wave * kq;
            float s = 0.f;
            for (int k = 0; k < kq; ++k) s = fmaf(actT[(k0 + k) * kMS + lane], ws[k0 + k], s);
            part[wave * 192 + lane] = s;
        }
        __syncthreads();
        if (t < 64) {
            const float sigma = part[t] + part[192 + t] + part[384 + t] + part[576 + t] + blob[L.scal];
            const int64_t n = n0 + t;
            const bool ok = n < N;
            if (!FUSED) {
                if (ok) A.out[((int64_t)b * N + n) * (F + 4) + F + 3] = sigma;
            } else {
                // ---- compositing weights for the 64 samples of this tile (volume_rendering.py:18-46)
                const int s_idx = (int)(n % S);
                const int64_t gi = (int64_t)b * N + n;
                float alpha = 0.f, f = 1.f, z = 0.f;
                if (ok) {
                    z = A.z_vals[gi];
                    const float delta = (s_idx == S - 1) ? 1e9f : A.z_vals[gi + 1] - z;
                    const float sg = sigma + (A.noise ? A.noise[gi] : 0.f);
                    alpha = 1.f - expf(-delta * density(sg, A.clamp_mode));
                    f = (1.f - alpha) + 1e-12f;
                }
                const int sl = t & (seglen - 1);
                float incl = f;
                for (int off = 1; off < seglen; off <<= 1) {
                    const float u = __shfl_up(incl, off, 64);
                    if (sl >= off) incl *= u;
                }
                float excl = __shfl_up(incl, 1, 64);
                if (sl == 0) excl = 1.f;
                float w = alpha * (carryT * excl);
                float wsum = w, dsum = w * z;
                for (int off = seglen >> 1; off > 0; off >>= 1) {
                    wsum += __shfl_xor(wsum, off, 64);
                    dsum += __shfl_xor(dsum, off, 64);
                }
                const int seg_last = (t | (seglen - 1));
                const float z_last = __shfl(z, seg_last, 64);
                carryT *= __shfl(incl, 63, 64);
                carryW += wsum;
                carryD += dsum;
                float bg = 0.f;
                if (last_tile) {
                    bg = 1.f - carryW;
                    if (ok && s_idx == S - 1) {
                        A.depth[gi / S] = carryD + bg * z_last;
                        if (A.last_back) w += bg;
                    }
                }
                if (ok) A.weights[gi] = w;
                wgt[t] = w;
                bgl[t] = bg;
            }
            sigv[t] = sigma;
        }

        // ---- colour branch: FiLM on [dir, x] with the last frequency/phase slice (modulated.py:67-68)
        zero_acc<NTW>(acc);
        if (A.dirs)
            gemm_phase<NTW>(acc, dirT, reinterpret_cast<const float4*>(blob + L.w_color), 2, 0, 2 + L.KBH, L.NT, wave, lane);
        gemm_phase<NTW>(acc, actT, reinterpret_cast<const float4*>(blob + L.w_color), L.KBH, 2, 2 + L.KBH, L.NT, wave, lane);
        __syncthreads();
        {
            const float* __restrict__ bb = blob + L.b_color;
            const float* __restrict__ wd = blob + L.w_dir + 2 * L.HdP;      // weight of dir.z
            const bool locked = A.dirs == nullptr;                           // dir = (0,0,-1)
            const int off = 3 * Hd;
            store_act<NTW>(acc, actT, L.NT, Hd, wave, lane,
                           [&](int n) { return Film{bb[n] - (locked ? wd[n] : 0.f), fr[off + n] * 15.f + 30.f, ph[off + n]}; },
                           [](float v, const Film& c) { return sin_accurate(fmaf(c.f, v + c.b, c.p)); });
        }
        __syncthreads();

        // ---- rgb head (VALU) and feature head (MFMA)
        {
            const float* __restrict__ wr = blob + L.w_rgb;
            const int kq = L.HdP / 4, k0 = wave * kq;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < kq; ++k) {
                const float x = actT[(k0 + k) * kMS + lane];
                s0 = fmaf(x, wr[k0 + k], s0);
                s1 = fmaf(x, wr[L.HdP + k0 + k], s1);
                s2 = fmaf(x, wr[2 * L.HdP + k0 + k], s2);
            }
            part[wave * 192 + lane] = s0;
            part[wave * 192 + 64 + lane] = s1;
            part[wave * 192 + 128 + lane] = s2;
        }
        f32x16 accF[2][NTW];
        zero_acc<NTW>(accF);
        gemm_phase<NTW>(accF, actT, reinterpret_cast<const float4*>(blob + L.w_feat), L.KBH, 0, L.KBH, L.NTF, wave, lane);
        __syncthreads();
        if (t < 192) {
            const int c = t >> 6, m = t & 63;
            const float v = part[c * 64 + m] + part[192 + c * 64 + m] + part[384 + c * 64 + m] + part[576 + c * 64 + m] +
                            blob[L.scal + 1 + c];
            const float rgb = 1.f / (1.f + expf(-v));
            const int64_t n = n0 + m;
            if (!FUSED) {
                if (n < N) A.out[((int64_t)b * N + n) * (F + 4) + c] = rgb;
            } else {
                rgbv[m * 3 + c] = rgb;
            }
        }
        if (!FUSED) {
            const float* __restrict__ bf = blob + L.b_feat;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int nt = wave + 4 * i;
                const int n = nt * 32 + j;
                if (nt >= L.NTF || n >= F) continue;
                const float bias = bf[n];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mt * 32 + (r >> 2) * 8 + 4 * h + (r & 3);
                        const int64_t pn = n0 + m;
                        if (pn < N) A.out[((int64_t)b * N + pn) * (F + 4) + 3 + n] = accF[mt][i][r] + bias;
                    }
            }
        } else {
            __syncthreads();       // rgbv visible
            const int C = F + 3;
            const int nseg = 64 / seglen;
            // colour channels: a few threads walk their ray's rows
            if (t < nseg * 3) {
                const int seg = t / 3, c = t - seg * 3;
                float s = 0.f;
                for (int q = 0; q < seglen; ++q) s = fmaf(wgt[seg * seglen + q], rgbv[(seg * seglen + q) * 3 + c], s);
                rgbacc += s;     // only meaningful for nseg == 1 (multi-tile rays); otherwise reset each tile
                const int64_t n_first = n0 + (int64_t)seg * seglen;
                if (last_tile && n_first < N) {
                    const int64_t ray = ((int64_t)b * N + n_first) / S;
                    const float tot = (S > 64 ? rgbacc : s) + (A.white_back ? bgl[seg * seglen] : 0.f);
                    A.feats[ray * C + c] = tot;
                }
            }
            // feature channels: weighted sum over the rows of each ray, straight from the accumulators
            const float* __restrict__ bf = blob + L.b_feat;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int nt = wave + 4 * i;
                const int n = nt * 32 + j;
                const bool okn = nt < L.NTF && n < F;
                const float bias = okn ? bf[n] : 0.f;
                float s8[2][4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wgt + mt * 32 + rg * 8 + 4 * h);
                        float s = (accF[mt][i][rg * 4 + 0] + bias) * w4.x;
                        s = fmaf(accF[mt][i][rg * 4 + 1] + bias, w4.y, s);
                        s = fmaf(accF[mt][i][rg * 4 + 2] + bias, w4.z, s);
                        s = fmaf(accF[mt][i][rg * 4 + 3] + bias, w4.w, s);
                        s += __shfl_xor(s, 32, 64);
                        s8[mt][rg] = s;
                    }
                if (S >= 64) {
                    rayacc[i] += ((s8[0][0] + s8[0][1]) + (s8[0][2] + s8[0][3])) + ((s8[1][0] + s8[1][1]) + (s8[1][2] + s8[1][3]));
                    if (last_tile && okn && h == 0 && n0 < N) {
                        const int64_t ray = ((int64_t)b * N + n0) / S;
                        A.feats[ray * C + 3 + n] = rayacc[i] + (A.white_back ? bgl[0] : 0.f);
                    }
                } else {
                    const int g8 = S >> 3;                       // 8-row groups per ray: 1, 2 or 4
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            if (rg % g8 != 0) continue;
                            float s = s8[mt][rg];
                            if (g8 >= 2) s += s8[mt][rg + 1 < 4 ? rg + 1 : 3];
                            if (g8 == 4) s += s8[mt][2] + s8[mt][3];
                            const int m_first = mt * 32 + rg * 8;
                            const int64_t n_first = n0 + m_first;
                            if (okn && h == 0 && n_first < N) {
                                const int64_t ray = ((int64_t)b * N + n_first) / S;
                                A.feats[ray * C + 3 + n] = s + (A.white_back ? bgl[m_first] : 0.f);
                            }
                        }
                }
            }
        }
        __syncthreads();     // actT / part / wgt are rewritten by the next tile
    }
}

size_t lds_bytes(const Layout& L) {
    const int act_rows = L.HdP > 40 ? L.HdP : 40;
    return sizeof(float) * ((size_t)act_rows * kMS + 16 * kMS + 4 * 3 * 64 + 64 + 64 + 64 * 3 + 64);
}

template <int NTW, bool FUSED>
int launch_one(const Args& A, int B, int64_t groups, hipStream_t st) {
    const size_t lds = lds_bytes(A.L);
    H3D_ALLOW_MAX_LDS((field_kernel<NTW, FUSED>));
    h3d::pre_launch();
    hipLaunchKernelGGL((field_kernel<NTW, FUSED>), dim3((unsigned)groups, (unsigned)B), dim3(kFieldThreads), lds, st, A);
    return h3d::launch_status(FUSED ? "h3d_render_fused" : "h3d_neural_field");
}

template <bool FUSED>
int launch(const Args& A, int B, int64_t groups, hipStream_t st) {
    const int widest = A.L.HdP > A.L.FP ? A.L.HdP : A.L.FP;
    const int ntw = (widest / 32 + 3) / 4;
    switch (ntw) {
        case 1: return launch_one<1, FUSED>(A, B, groups, st);
        case 2: return launch_one<2, FUSED>(A, B, groups, st);
        case 3: return launch_one<3, FUSED>(A, B, groups, st);
        case 4: return launch_one<4, FUSED>(A, B, groups, st);
        default:
            h3d::set_error("neural field: hidden/feature width %d exceeds the 512 this build supports", widest);
            return H3D_EUNSUPPORTED;
    }
}

int check_common(const void* packed, const float* points, const float* geo, const float* freq, const float* phase,
                 int B, int64_t N, int Hd, int F, int geo_stride) {
    H3D_REQUIRE(packed && points && geo && freq && phase, "neural field: null pointer");
    H3D_REQUIRE(h3d::aligned16(packed), "neural field: packed weights must be 16-byte aligned");
    H3D_REQUIRE(B >= 0 && B <= 65535 && N >= 0, "neural field: bad B=%d N=%lld", B, (long long)N);
    H3D_REQUIRE(Hd >= 1 && F >= 1, "neural field: bad widths Hd=%d F=%d", Hd, F);
    H3D_REQUIRE(geo_stride >= 31, "neural field: geo_stride=%d must be >= 31", geo_stride);
    return H3D_OK;
}

}  // namespace

extern "C" int64_t h3d_field_pack_size(int Hd, int F) {
    if (Hd < 1 || F < 1) return -1;
    return make_layout(Hd, F).total * (int64_t)sizeof(float);
}

extern "C" int h3d_field_pack(const h3d_field_params* p, int Hd, int F, void* blob_) {
    H3D_REQUIRE(p && blob_, "h3d_field_pack: null pointer");
    H3D_REQUIRE(Hd >= 1 && F >= 1, "h3d_field_pack: bad widths");
    const Layout L = make_layout(Hd, F);
    float* blob = static_cast<float*>(blob_);
    for (int64_t i = 0; i < L.total; ++i) blob[i] = 0.f;
    pack_matrix(p->w_coord, 3, 0, 3, Hd, 2, L.NT, blob + L.w_coord);
    pack_matrix(p->w_geo, 31, 0, 31, Hd, 4, L.NT, blob + L.w_geo);
    // FiLM 0 consumes [coord features (Hd) | geometry features (Hd)]: two K ranges of HdP rows each
    {
        float* dst = blob + L.w_film[0];
        const int KB2 = 2 * L.KBH;
        for (int nt = 0; nt < L.NT; ++nt)
            for (int kb = 0; kb < KB2; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int half = kb >= L.KBH;
                        const int k = 8 * (kb - half * L.KBH) + 4 * (lane >> 5) + e;
                        const int n = 32 * nt + (lane & 31);
                        float v = 0.f;
                        if (k < Hd && n < Hd) v = p->w_film[0][(int64_t)n * 2 * Hd + half * Hd + k];
                        dst[(((int64_t)nt * KB2 + kb) * 64 + lane) * 4 + e] = v;
                    }
    }
    for (int l = 1; l < 4; ++l) pack_matrix(p->w_film[l], Hd, 0, Hd, Hd, L.KBH, L.NT, blob + L.w_film[l]);
    // colour layer input = [dir (3) | x (Hd)]: k-block 0 holds the direction rows, blocks 1.. the x rows
    {
        float* dst = blob + L.w_color;
        const int KBt = 2 + L.KBH;
        for (int nt = 0; nt < L.NT; ++nt)
            for (int kb = 0; kb < KBt; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int kk = 4 * (lane >> 5) + e;
                        const int n = 32 * nt + (lane & 31);
                        float v = 0.f;
                        if (n < Hd) {
                            if (kb == 0) { if (kk < 3) v = p->w_color[(int64_t)n * (Hd + 3) + kk]; }
                            else if (kb >= 2) { const int k = 8 * (kb - 2) + kk; if (k < Hd) v = p->w_color[(int64_t)n * (Hd + 3) + 3 + k]; }
                        }
                        dst[(((int64_t)nt * KBt + kb) * 64 + lane) * 4 + e] = v;
                    }
    }
    pack_matrix(p->w_feat, Hd, 0, Hd, F, L.KBH, L.NTF, blob + L.w_feat);
    for (int n = 0; n < Hd; ++n) {
        blob[L.b_coord + n] = p->b_coord[n];
        blob[L.b_geo + n] = p->b_geo[n];
        for (int l = 0; l < 4; ++l) blob[L.b_film[l] + n] = p->b_film[l][n];
        blob[L.b_color + n] = p->b_color[n];
        for (int c = 0; c < 3; ++c) {
            blob[L.w_dir + c * L.HdP + n] = p->w_color[(int64_t)n * (Hd + 3) + c];
            blob[L.w_rgb + c * L.HdP + n] = p->w_rgb[(int64_t)c * Hd + n];
        }
        blob[L.w_sigma + n] = p->w_sigma[n];
    }
    for (int n = 0; n < F; ++n) blob[L.b_feat + n] = p->b_feat[n];
    blob[L.scal] = p->b_sigma[0];
    for (int c = 0; c < 3; ++c) blob[L.scal + 1 + c] = p->b_rgb[c];
    return H3D_OK;
}

extern "C" int h3d_neural_field(const void* packed, const float* points, const float* geo, const float* dirs,
                                const float* freq, const float* phase, float* out, int B, int64_t N, int Hd, int F,
                                int geo_stride, float input_scaler, h3d_stream_t stream) {
    int rc = check_common(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(out, "h3d_neural_field: null output");
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const float*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase; A.out = out;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = 64; A.input_scaler = input_scaler;
    A.L = make_layout(Hd, F);
    const int64_t groups = (N + 63) / 64;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_neural_field: N too large");
    return launch<false>(A, B, groups, static_cast<hipStream_t>(stream));
}

extern "C" int h3d_render_fused(const void* packed, const float* points, const float* geo, const float* dirs,
                                const float* freq, const float* phase, const float* z_vals, const float* noise,
                                float* feats, float* depth, float* weights, int B, int R, int S, int Hd, int F,
                                int geo_stride, float input_scaler, int clamp_mode, int last_back, int white_back,
                                h3d_stream_t stream) {
    const int64_t N = (int64_t)R * S;
    int rc = check_common(packed, points, geo, freq, phase, B, N, Hd, F, geo_stride);
    if (rc) return rc;
    H3D_REQUIRE(z_vals && feats && depth && weights, "h3d_render_fused: null pointer");
    H3D_REQUIRE(clamp_mode == 0 || clamp_mode == 1, "h3d_render_fused: clamp_mode must be 0 (relu) or 1 (softplus)");
    H3D_REQUIRE(R >= 0 && S >= 1, "h3d_render_fused: bad R=%d S=%d", R, S);
    const bool ok_s = (S >= 8 && S <= 64 && (S & (S - 1)) == 0) || (S > 64 && S % 64 == 0);
    if (!ok_s) {
        h3d::set_error("h3d_render_fused: S=%d unsupported by the fused kernel (needs 8,16,32,64 or a multiple of 64); "
                       "use h3d_neural_field + h3d_ray_integrate", S);
        return H3D_EUNSUPPORTED;
    }
    if (B == 0 || N == 0) return H3D_OK;
    Args A{};
    A.blob = static_cast<const float*>(packed);
    A.points = points; A.geo = geo; A.dirs = dirs; A.freq = freq; A.phase = phase;
    A.z_vals = z_vals; A.noise = noise; A.feats = feats; A.depth = depth; A.weights = weights;
    A.N = N; A.Hd = Hd; A.F = F; A.geo_stride = geo_stride; A.S = S; A.input_scaler = input_scaler;
    A.clamp_mode = clamp_mode; A.last_back = last_back; A.white_back = white_back;
    A.L = make_layout(Hd, F);
    const int group = S > 64 ? S : 64;
    const int64_t groups = (N + group - 1) / group;
    H3D_REQUIRE(groups < (int64_t(1) << 31), "h3d_render_fused: too many rays");
    return launch<true>(A, B, groups, static_cast<hipStream_t>(stream));
}
