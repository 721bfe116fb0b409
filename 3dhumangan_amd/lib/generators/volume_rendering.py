"""Volume rendering entry points with the reference's names and argument meaning
(reference: lib/generators/volume_rendering.py), backed by HIP kernels in libh3d.so.

Differences that are deliberate:
  * every random tensor can be *injected* (``noise=`` / ``jitter=``); when it is not, it is drawn with
    torch on the device exactly where the reference draws it, so RNG consumption order is unchanged;
  * tensors must live on a ROCm device -- there is no CPU path in the product.
"""
import torch

from ... import _lib

_CLAMP = {"relu": 0, "softplus": 1}


_need_cuda = _lib.need_cuda


def sample_rays(focals, scales, cam2world_matrix, num_steps, resolution, ray_start, ray_end, jitter=None,
                perturb=True):
    """Fused ray set-up: the reference's get_initial_rays_weak_perspective (volume_rendering.py:86-110) +
    perturb_points (:124-130) + the camera->world transform of transform_sampled_points (:133-170).

    resolution = (W, H) as in the reference.  ``jitter`` is the U(0,1) tensor [B,R,S,1] the reference draws
    at :126; when ``perturb`` and it is None it is drawn here (same shape, same place in the RNG stream).
    -> points [B,R*S,3] (world), z_vals [B,R,S,1]."""
    W, H = resolution
    _need_cuda(focals, scales, cam2world_matrix, jitter)
    B = focals.shape[0]
    R = W * H
    dev = focals.device
    if perturb and jitter is None:
        jitter = torch.rand((B, R, num_steps, 1), device=dev)
    jt = None if jitter is None else jitter.reshape(B, R, num_steps).contiguous().float()
    points = torch.empty((B, R * num_steps, 3), device=dev, dtype=torch.float32)
    z_vals = torch.empty((B, R, num_steps, 1), device=dev, dtype=torch.float32)
    f32 = focals.contiguous().float()
    s32 = scales.contiguous().float()
    c2w = cam2world_matrix.contiguous().float()
    rc = _lib.load().h3d_ray_setup(_lib.ptr(f32), _lib.ptr(s32), _lib.ptr(c2w), _lib.ptr(jt), _lib.ptr(points),
                                   _lib.ptr(z_vals), B, H, W, num_steps, float(ray_start), float(ray_end),
                                   _lib.stream_handle())
    _lib.check(rc, "h3d_ray_setup")
    return points, z_vals


def ray_directions_world(focals, cam2world_matrix, resolution, num_steps):
    """World-space view directions per sample [B,R*S,3] (reference volume_rendering.py:93-101, 113-121, 157-160).
    Only needed when lock_view_dependence is off; a few small torch ops on the device."""
    W, H = resolution
    B, dev = focals.shape[0], focals.device
    xs = torch.linspace(-W / H, W / H, W, device=dev).repeat(H)
    ys = torch.linspace(-1, 1, H, device=dev).repeat_interleave(W)
    d = torch.stack([xs.expand(B, -1), ys.expand(B, -1), focals.float()[:, None].expand(B, W * H)], dim=-1)
    d = d / (torch.norm(d, dim=-1, keepdim=True) + 1e-12)
    d = torch.bmm(cam2world_matrix[:, :3, :3].float(), d.transpose(1, 2)).transpose(1, 2)
    return d.unsqueeze(2).expand(B, W * H, num_steps, 3).reshape(B, W * H * num_steps, 3).contiguous()


def ray_integration(input, z_vals, device=None, noise_std=0.5, last_back=False, white_back=False,
                    clamp_mode=None, fill_mode=None, noise=None):
    """NeRF volume integration.  reference: volume_rendering.py:12-56.

    input [B,R,S,C+1] (density last), z_vals [B,R,S,1] -> (features [B,R,C], depth [B,R,1], weights [B,R,S,1]).
    ``noise`` (already scaled, [B,R,S,1]) overrides the internal ``randn * noise_std`` draw."""
    if clamp_mode not in _CLAMP:
        raise Exception("Need to choose clamp mode")
    if fill_mode is not None:
        raise NotImplementedError("fill_mode is a debug visualisation of the reference and is not provided")
    _need_cuda(input, z_vals, noise)
    B, R, S, C1 = input.shape
    field = input.contiguous().float()
    z = z_vals.reshape(B, R, S).contiguous().float()
    if noise is None:
        # the reference always consumes RNG here (volume_rendering.py:24), even for noise_std == 0
        noise = torch.randn((B, R, S, 1), device=field.device) * noise_std
        if noise_std == 0:
            noise = None
    nz = None if noise is None else noise.reshape(B, R, S).contiguous().float()
    feats = torch.empty((B, R, C1 - 1), device=field.device, dtype=torch.float32)
    depth = torch.empty((B, R, 1), device=field.device, dtype=torch.float32)
    weights = torch.empty((B, R, S, 1), device=field.device, dtype=torch.float32)
    lib = _lib.load()
    rc = lib.h3d_ray_integrate(_lib.ptr(field), _lib.ptr(z), _lib.ptr(nz), _lib.ptr(feats), _lib.ptr(depth),
                               _lib.ptr(weights), B * R, S, C1 - 1, _CLAMP[clamp_mode], int(bool(last_back)),
                               int(bool(white_back)), _lib.stream_handle())
    _lib.check(rc, "h3d_ray_integrate")
    return feats, depth, weights
