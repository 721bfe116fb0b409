// A3 ray set-up for gfx950: one thread per sample point.
// Reference semantics: lib/generators/volume_rendering.py:86-110 (pixel grid, weak-perspective rays, depth
// planes), :124-130 (stratified jitter applied to depth and point), :133-170 (camera -> world).
#include "common.hpp"

namespace {

// torch.linspace(start, end, n)[i] in fp32 (symmetric evaluation, as ATen does).
__device__ __forceinline__ float linspace_at(float start, float end, int n, int i) {
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return (i < n / 2) ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

__global__ __launch_bounds__(256) void ray_setup_kernel(const float* __restrict__ focals,
                                                        const float* __restrict__ scales,
                                                        const float* __restrict__ cam2world,
                                                        const float* __restrict__ jitter,
                                                        float* __restrict__ points, float* __restrict__ z_vals,
                                                        int render_h, int render_w, int S, float ray_start,
                                                        float ray_end, int64_t per_image) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_image) return;
    const int s = (int)(i % S);
    const int r = (int)(i / S);
    const int ix = r % render_w, iy = r / render_w;
    const float focal = focals[b];
    const float span = (float)render_w / (float)render_h;
    const float x = linspace_at(-span, span, render_w, ix);
    const float y = linspace_at(-1.f, 1.f, render_h, iy);
    const float inv = 1.f / (sqrtf(x * x + y * y + focal * focal) + 1e-12f);
    const float dx = x * inv, dy = y * inv, dz = focal * inv;
    const float zc = focal / scales[b];
    float z = linspace_at(ray_start, ray_end, S, s) + zc;
    float px = dx * z, py = dy * z, pz = dz * z;
    if (jitter) {
        const float z0 = linspace_at(ray_start, ray_end, S, 0) + zc;
        const float z1 = (S > 1) ? linspace_at(ray_start, ray_end, S, 1) + zc : z0;
        const float off = (jitter[(int64_t)b * per_image + i] - 0.5f) * (z1 - z0);
        z += off;
        px += off * dx; py += off * dy; pz += off * dz;
    }
    const float* __restrict__ M = cam2world + b * 16;
    float* o = points + ((int64_t)b * per_image + i) * 3;
    o[0] = M[0] * px + M[1] * py + M[2] * pz + M[3];
    o[1] = M[4] * px + M[5] * py + M[6] * pz + M[7];
    o[2] = M[8] * px + M[9] * py + M[10] * pz + M[11];
    z_vals[(int64_t)b * per_image + i] = z;
}

}  // namespace

extern "C" int h3d_ray_setup(const float* focals, const float* scales, const float* cam2world, const float* jitter,
                             float* points, float* z_vals, int B, int render_h, int render_w, int S, float ray_start,
                             float ray_end, h3d_stream_t stream) {
    H3D_REQUIRE(focals && scales && cam2world && points && z_vals, "h3d_ray_setup: null pointer");
    H3D_REQUIRE(B >= 0 && B <= 65535, "h3d_ray_setup: B=%d out of range", B);
    H3D_REQUIRE(render_h >= 1 && render_w >= 1 && S >= 1, "h3d_ray_setup: bad geometry %dx%dx%d", render_h, render_w, S);
    if (B == 0) return H3D_OK;
    const int64_t per_image = (int64_t)render_h * render_w * S;
    const int64_t gx = (per_image + 255) / 256;
    H3D_REQUIRE(gx < (int64_t(1) << 31), "h3d_ray_setup: too many points per image");
    h3d::pre_launch();
    hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), focals,
                       scales, cam2world, jitter, points, z_vals, render_h, render_w, S, ray_start, ray_end, per_image);
    return h3d::launch_status("h3d_ray_setup");
}
