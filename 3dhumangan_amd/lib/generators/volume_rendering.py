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


def _integrate(field, z, nz, clamp, last_back, white_back):
    B, R, S, C1 = field.shape
    feats = torch.empty((B, R, C1 - 1), device=field.device, dtype=torch.float32)
    depth = torch.empty((B, R, 1), device=field.device, dtype=torch.float32)
    weights = torch.empty((B, R, S, 1), device=field.device, dtype=torch.float32)
    rc = _lib.load().h3d_ray_integrate(_lib.ptr(field), _lib.ptr(z), _lib.ptr(nz), _lib.ptr(feats), _lib.ptr(depth),
                                       _lib.ptr(weights), B * R, S, C1 - 1, clamp, last_back, white_back,
                                       _lib.stream_handle())
    _lib.check(rc, "h3d_ray_integrate")
    return feats, depth, weights


class _RayIntegration(torch.autograd.Function):
    """h3d_ray_integrate with its hand-written adjoint h3d_ray_integrate_bwd (gradient w.r.t. the field tensor; the sample
    depths and the noise are not learnable on this path and get none)."""

    @staticmethod
    def forward(ctx, field, z, nz, clamp, last_back, white_back):
        ctx.save_for_backward(field, z, nz)
        ctx.flags = (clamp, last_back, white_back)
        return _integrate(field, z, nz, clamp, last_back, white_back)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_feats, g_depth, g_weights):
        field, z, nz = ctx.saved_tensors
        clamp, last_back, white_back = ctx.flags
        B, R, S, C1 = field.shape
        gf = (torch.zeros((B, R, C1 - 1), device=field.device) if g_feats is None else g_feats.contiguous().float())
        gd = None if g_depth is None else g_depth.contiguous().float()
        gw = None if g_weights is None else g_weights.contiguous().float()
        d_field = torch.empty_like(field)
        rc = _lib.load().h3d_ray_integrate_bwd(_lib.ptr(field), _lib.ptr(z), _lib.ptr(nz), _lib.ptr(gf), _lib.ptr(gd),
                                               _lib.ptr(gw), _lib.ptr(d_field), B * R, S, C1 - 1, clamp, last_back,
                                               white_back, _lib.stream_handle())
        _lib.check(rc, "h3d_ray_integrate_bwd")
        return d_field, None, None, None, None, None


def ray_integration(input, z_vals, device=None, noise_std=0.5, last_back=False, white_back=False,
                    clamp_mode=None, fill_mode=None, noise=None, consume_rng=True):
    """NeRF volume integration.  reference: volume_rendering.py:12-56.

    input [B,R,S,C+1] (density last), z_vals [B,R,S,1] -> (features [B,R,C], depth [B,R,1], weights [B,R,S,1]).
    ``noise`` (already scaled, [B,R,S,1]) overrides the internal ``randn * noise_std`` draw; ``consume_rng=False``: the caller
    has already drawn (and possibly discarded) the noise tensor, nothing is drawn here.  Differentiable w.r.t. ``input``
    (h3d_ray_integrate_bwd)."""
    if clamp_mode not in _CLAMP:
        raise Exception("Need to choose clamp mode")
    if fill_mode is not None:
        raise NotImplementedError("fill_mode is a debug visualisation of the reference and is not provided")
    _need_cuda(input, z_vals, noise)
    B, R, S, C1 = input.shape
    field = input.contiguous().float()
    z = z_vals.reshape(B, R, S).contiguous().float()
    if noise is None and consume_rng:
        # the reference always consumes RNG here (volume_rendering.py:24), even for noise_std == 0
        noise = torch.randn((B, R, S, 1), device=field.device) * noise_std
        if noise_std == 0:
            noise = None
    nz = None if noise is None else noise.reshape(B, R, S).contiguous().float()
    flags = (_CLAMP[clamp_mode], int(bool(last_back)), int(bool(white_back)))
    if torch.is_grad_enabled() and field.requires_grad:
        return _RayIntegration.apply(field, z.detach(), None if nz is None else nz.detach(), *flags)
    return _integrate(field, z, nz, *flags)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """Inverse-CDF importance sampling, reference volume_rendering.py:261-303 (same arguments).  ``u`` injects the
    uniform draws [N_rays, N_importance]; otherwise they are drawn here exactly where the reference draws them (:285),
    or taken as linspace(0, 1) when ``det``.  bins [N_rays, n+1], weights [N_rays, n] -> samples [N_rays, N_importance]."""
    _need_cuda(bins, weights, u)
    n_rays, n = weights.shape
    if u is None:
        u = (torch.linspace(0, 1, N_importance, device=bins.device).expand(n_rays, N_importance) if det
             else torch.rand(n_rays, N_importance, device=bins.device))
    b32, w32, u32 = bins.contiguous().float(), weights.contiguous().float(), u.contiguous().float()
    out = torch.empty((n_rays, N_importance), device=bins.device, dtype=torch.float32)
    rc = _lib.load().h3d_sample_pdf(_lib.ptr(b32), _lib.ptr(w32), _lib.ptr(u32), _lib.ptr(out), n_rays, n + 1, N_importance,
                                    float(eps), _lib.stream_handle())
    _lib.check(rc, "h3d_sample_pdf")
    return out


def ray_frame_world(focals, cam2world_matrix, resolution):
    """World-space ray origins [B,3] and unit directions [B,R,3] (transformed_ray_origins / transformed_ray_directions of
    volume_rendering.py:133-170); a few small torch ops on the device."""
    W, H = resolution
    d = ray_directions_world(focals, cam2world_matrix, resolution, 1)
    return cam2world_matrix[:, :3, 3].float().contiguous(), d


def ray_points(origins, dirs, z_vals):
    """origins [B,3], dirs [B,R,3], z_vals [B,R,S,1] -> points [B,R*S,3] = origin + dir * z (map3d_generator.py:464-466)."""
    _need_cuda(origins, dirs, z_vals)
    B, R, S = z_vals.shape[0], z_vals.shape[1], z_vals.shape[2]
    o, d, z = origins.contiguous().float(), dirs.contiguous().float(), z_vals.contiguous().float()
    pts = torch.empty((B, R * S, 3), device=z.device, dtype=torch.float32)
    rc = _lib.load().h3d_ray_points(_lib.ptr(o), _lib.ptr(d), _lib.ptr(z), _lib.ptr(pts), B, R, S, _lib.stream_handle())
    _lib.check(rc, "h3d_ray_points")
    return pts


def merge_samples(fine_output, coarse_output, fine_z_vals, z_vals):
    """cat([fine, coarse]) sorted by depth along the sample axis (map3d_generator.py:504-509).
    fine/coarse_output [B,R,S*,C+1], fine_z_vals / z_vals [B,R,S*,1] -> all_outputs [B,R,Sf+Sc,C+1], all_z_vals [B,R,Sf+Sc,1]."""
    _need_cuda(fine_output, coarse_output, fine_z_vals, z_vals)
    B, R, Sf, C1 = fine_output.shape
    Sc = coarse_output.shape[2]
    f, c = fine_output.contiguous().float(), coarse_output.contiguous().float()
    fz, cz = fine_z_vals.contiguous().float(), z_vals.contiguous().float()
    out = torch.empty((B, R, Sf + Sc, C1), device=f.device, dtype=torch.float32)
    out_z = torch.empty((B, R, Sf + Sc, 1), device=f.device, dtype=torch.float32)
    rc = _lib.load().h3d_merge_samples(_lib.ptr(f), _lib.ptr(c), _lib.ptr(fz), _lib.ptr(cz), _lib.ptr(out), _lib.ptr(out_z),
                                       B * R, Sf, Sc, C1, _lib.stream_handle())
    _lib.check(rc, "h3d_merge_samples")
    return out, out_z
