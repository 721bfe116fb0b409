"""CPU oracle for the 3DHumanGAN generator forward pass  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) functional restatement of the
reference algorithm for the hot path named by BASELINE.json.  It exists so that
the HIP kernels can be checked against something that (a) has been pinned to
the real reference (tests/golden/*.npz were produced by importing
/root/reference in the build container, see tests/golden/make_golden.py) and
(b) can travel to the GPU box.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product package
(3dhumangan_amd) never does, and fails loudly if libh3d.so is missing.

Every function works on a flat ``state`` dict that uses the reference's
state_dict key schema and takes every random tensor the
reference would have drawn as an explicit argument.

Parity status: pinned against the imported reference for A1-A11 and the
plugin-op ``_ref`` paths.  The K=1 nearest-vertex search is pytorch3d 0.6.2
``knn_points`` in the reference (lib/components/smpl.py:220), which is not
vendored and not installed: that boundary is "parity unpinned" (only tie
breaking can differ; documented in DESIGN.md).

Citations are file:line under /root/reference.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# small helpers


def second_moment_normalize(x, dim=1, eps=1e-8):
    """lib/components/util.py:61-62."""
    return x * torch.rsqrt(x.square().mean(dim=dim, keepdim=True) + eps)


def _lin(state, prefix, x):
    return F.linear(x, state[prefix + ".weight"].to(x.dtype), state[prefix + ".bias"].to(x.dtype))


# --------------------------------------------------------------------------
# A1: FiLM mapping network  (lib/components/mapping_networks.py:15-41)


def film_mapping(state, z, prefix="neural_field_mapping_network"):
    h = second_moment_normalize(z.to(torch.float32) if z.dtype != torch.float64 else z)
    for i in (0, 2, 4):
        h = F.leaky_relu(_lin(state, f"{prefix}.network.{i}", h), 0.2)
    out = _lin(state, f"{prefix}.network.6", h)
    half = out.shape[-1] // 2
    return out[..., :half], out[..., half:]


# --------------------------------------------------------------------------
# P1: bias_act reference semantics (lib/components/ops/bias_act.py:91-120,
# activation table :20-31; kernel spec lib/components/ops/bias_act.cu:23-147)

_SQRT2 = math.sqrt(2.0)
ACTIVATIONS = {
    # name: (fn(x, alpha), default alpha, default gain, cuda_idx)
    "linear": (lambda x, a: x, 0.0, 1.0, 1),
    "relu": (lambda x, a: torch.relu(x), 0.0, _SQRT2, 2),
    "lrelu": (lambda x, a: F.leaky_relu(x, a), 0.2, _SQRT2, 3),
    "tanh": (lambda x, a: torch.tanh(x), 0.0, 1.0, 4),
    "sigmoid": (lambda x, a: torch.sigmoid(x), 0.0, 1.0, 5),
    "elu": (lambda x, a: F.elu(x), 0.0, 1.0, 6),
    "selu": (lambda x, a: F.selu(x), 0.0, 1.0, 7),
    "softplus": (lambda x, a: F.softplus(x), 0.0, 1.0, 8),
    "swish": (lambda x, a: torch.sigmoid(x) * x, 0.0, _SQRT2, 9),
}


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    fn, def_alpha, def_gain, _ = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


# --------------------------------------------------------------------------
# A2: StyleGAN-style two-branch mapping network
# (lib/components/mapping_networks.py:92-121 FC layer, :124-216 network)


def _fc(state, prefix, x, act, lr_mul, extra_weight_gain=1.0):
    w = state[prefix + ".weight"].to(x.dtype)
    b = state[prefix + ".bias"].to(x.dtype)
    w = w * (lr_mul / math.sqrt(w.shape[1]) * extra_weight_gain)
    b = b * lr_mul
    if act == "linear":
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return bias_act(x.matmul(w.t()), b, act=act)


def style_mapping(state, z, prefix="synthesis_mapping_network", trunk_layers=7, lr_mul=0.01):
    """Returns (implicit [B,1], styles [B,1,F])."""
    h = second_moment_normalize(z.to(torch.float32) if z.dtype != torch.float64 else z)
    for i in range(trunk_layers):
        h = _fc(state, f"{prefix}.trunk{i}", h, "lrelu", lr_mul)
    implicit = _fc(state, f"{prefix}.implicit0", h, "linear", lr_mul, extra_weight_gain=0.2)
    styles = _fc(state, f"{prefix}.superres0", h, "lrelu", lr_mul)
    return implicit, styles.unsqueeze(1)


# --------------------------------------------------------------------------
# A3: ray set-up (lib/generators/volume_rendering.py:86-110, 124-130, 133-170;
# lock-view override lib/generators/map3d_generator.py:417-420)


def ray_setup(focals, scales, cam2world, render_h, render_w, num_steps, ray_start, ray_end,
              jitter=None, lock_view_dependence=True, ray_subset=None):
    """jitter: U(0,1) tensor [B,R,S,1] (the reference draws it at
    volume_rendering.py:126) or None for no perturbation.
    ray_subset: optional LongTensor of ray indices r = y*render_w + x; only those rays are set up (every ray is
    independent of the others, so this is the full computation restricted to rows of the result; the jitter tensor is
    still the full-size one and is indexed here).  R below is then len(ray_subset).
    Returns points [B,R*S,3] (world), z_vals [B,R,S,1], dirs [B,R*S,3]."""
    B = focals.shape[0]
    dt = focals.dtype
    R = render_h * render_w
    span = render_w / render_h
    xs = torch.linspace(-span, span, render_w, dtype=dt)
    ys = torch.linspace(-1, 1, render_h, dtype=dt)
    # row-major over (y, x): pixel r = y*W + x
    px = xs.repeat(render_h)                      # x varies fastest
    py = ys.repeat_interleave(render_w)
    if ray_subset is not None:
        px, py, R = px[ray_subset], py[ray_subset], int(ray_subset.numel())
        if jitter is not None:
            jitter = jitter[:, ray_subset]
    d = torch.stack([px.expand(B, R), py.expand(B, R), focals[:, None].expand(B, R)], dim=-1)
    d = d / (torch.norm(d, dim=-1, keepdim=True) + 1e-12)
    z = torch.linspace(ray_start, ray_end, num_steps, dtype=dt).view(1, 1, num_steps, 1)
    z = z.expand(B, R, num_steps, 1) + (focals / scales).view(B, 1, 1, 1)
    pts = d.unsqueeze(2) * z
    if jitter is not None:
        off = (jitter - 0.5) * (z[:, :, 1:2] - z[:, :, 0:1])
        z = z + off
        pts = pts + off * d.unsqueeze(2)
    hom = F.pad(pts, (0, 1), value=1.0).reshape(B, R * num_steps, 4)
    world = torch.bmm(cam2world.to(dt), hom.transpose(1, 2)).transpose(1, 2)[..., :3]
    wd = torch.bmm(cam2world[:, :3, :3].to(dt), d.transpose(1, 2)).transpose(1, 2)
    dirs = wd.unsqueeze(2).expand(B, R, num_steps, 3).reshape(B, R * num_steps, 3)
    if lock_view_dependence:
        dirs = torch.zeros_like(dirs)
        dirs[..., 2] = -1
    return world.contiguous(), z.contiguous(), dirs.contiguous()


# --------------------------------------------------------------------------
# A4: SMPL geometry features (lib/components/smpl.py:210-249)


def nearest_vertex(points, vertices, chunk=4096):
    """K=1 nearest vertex; squared distance accumulated as (dx*dx+dy*dy)+dz*dz,
    first index wins ties.  Stand-in for pytorch3d.ops.knn_points (unpinned)."""
    B, N, _ = points.shape
    idx = torch.empty(B, N, dtype=torch.long)
    d2o = torch.empty(B, N, dtype=points.dtype)
    for b in range(B):
        v = vertices[b]
        for s in range(0, N, chunk):
            p = points[b, s:s + chunk]
            dx = p[:, None, 0] - v[None, :, 0]
            dy = p[:, None, 1] - v[None, :, 1]
            dz = p[:, None, 2] - v[None, :, 2]
            d2 = (dx * dx + dy * dy) + dz * dz
            m = d2.min(dim=1, keepdim=True).values
            ar = torch.arange(v.shape[0]).expand_as(d2)
            first = torch.where(d2 == m, ar, torch.full_like(ar, v.shape[0])).min(dim=1).values
            idx[b, s:s + chunk] = first
            d2o[b, s:s + chunk] = m[:, 0]
    return d2o, idx


def vertex_inverse_transforms(fk_matrices, lbs_weights):
    """[B,V,4,4] blended inverse bone transforms (smpl.py:217-218)."""
    ik = torch.inverse(fk_matrices.float()).to(lbs_weights.dtype)
    return torch.einsum("bvj,bjkl->bvkl", lbs_weights, ik)


def geo_features(points, skeletons, vertices, tpose_vertices, fk_matrices, lbs_weights,
                 legacy_mode=False, return_index=False):
    B, N, _ = points.shape
    joint_d = torch.cdist(points, skeletons) / 2.4
    vik = vertex_inverse_transforms(fk_matrices, lbs_weights)
    d2, idx = nearest_vertex(points.float(), vertices.float())
    M = torch.gather(vik.reshape(B, -1, 16), 1, idx[..., None].expand(B, N, 16)).reshape(B, N, 4, 4)
    hom = F.pad(points, (0, 1), value=1.0)
    cano = torch.einsum("bnij,bnj->bni", M, hom)[..., :3]
    cano = torch.stack([cano[..., 0] / 2.0, (cano[..., 1] + 0.2) / 2.0, cano[..., 2] / 1.3], dim=-1)
    tv = torch.gather(tpose_vertices, 1, idx[..., None].expand(B, N, 3))
    tv = torch.stack([tv[..., 0], tv[..., 1], tv[..., 2] / 0.2], dim=-1)
    nd = (torch.sqrt(d2) / 1.3).unsqueeze(-1).to(points.dtype)
    parts = [joint_d, cano, tv, nd] if legacy_mode else [cano, joint_d, tv, nd]
    out = torch.cat(parts, dim=-1)
    return (out, idx, d2) if return_index else out


# --------------------------------------------------------------------------
# A5: pose-conditioned FiLM-SIREN (lib/implicit_funcitions/modulated.py:41-75,
# lib/components/pigan_layers.py:63-87)


def neural_field(state, points, freq, phase, geo, dirs, input_scaler=1.0, prefix="neural_field",
                 num_blocks=4):
    """-> [B,N,F+4] with channel order [rgb(3), feat(F), sigma(1)]."""
    hd = state[f"{prefix}.sigma_layer.weight"].shape[1]
    f = (freq * 15 + 30).unsqueeze(1)
    ph = phase.unsqueeze(1)
    a = torch.sin(30.0 * _lin(state, f"{prefix}.first_layer_coord.layer", points * input_scaler))
    g = torch.sin(30.0 * _lin(state, f"{prefix}.first_layer_mod.layer", geo))
    x = torch.cat([a, g], dim=-1)
    for k in range(num_blocks):
        sl = slice(k * hd, (k + 1) * hd)
        x = torch.sin(f[..., sl] * _lin(state, f"{prefix}.network.{k}.layer", x) + ph[..., sl])
    sigma = _lin(state, f"{prefix}.sigma_layer", x)
    c = torch.cat([dirs, x], dim=-1)
    c = torch.sin(f[..., -hd:] * _lin(state, f"{prefix}.color_layer_sine.layer", c) + ph[..., -hd:])
    rgb = torch.sigmoid(_lin(state, f"{prefix}.color_layer_linear", c))
    feat = _lin(state, f"{prefix}.feature_layer_linear", c)
    return torch.cat([rgb, feat, sigma], dim=-1)


# --------------------------------------------------------------------------
# A6: volume integration (lib/generators/volume_rendering.py:12-56)


def ray_integration(field, z_vals, noise=None, clamp_mode="relu", last_back=False, white_back=False):
    """field [B,R,S,C+1] (sigma last), z_vals [B,R,S,1], noise [B,R,S,1] already
    scaled by noise_std (or None).  -> features [B,R,C], depth [B,R,1], weights [B,R,S,1]."""
    feats, sigma = field[..., :-1], field[..., -1:]
    delta = torch.cat([z_vals[:, :, 1:] - z_vals[:, :, :-1],
                       torch.full_like(z_vals[:, :, :1], 1e9)], dim=2)
    s = sigma if noise is None else sigma + noise
    dens = F.softplus(s) if clamp_mode == "softplus" else torch.relu(s)
    if clamp_mode not in ("softplus", "relu"):
        raise ValueError("Need to choose clamp mode")
    alpha = 1 - torch.exp(-delta * dens)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-12], dim=2), dim=2)[:, :, :-1]
    w = alpha * trans
    wsum = w.sum(2)
    w_last = w.clone()
    w_last[:, :, -1] = w_last[:, :, -1] + (1 - wsum)
    if last_back:
        w = w_last
        out = (w * feats).sum(2)
    else:
        out = (w * feats).sum(2)
    depth = (w_last * z_vals).sum(2)
    if white_back:
        out = out + 1 - wsum
    return out, depth, w


# --------------------------------------------------------------------------
# A7: bilinear resize, align_corners=False (map3d_generator.py:244-245)


def bilinear_resize(x, out_h, out_w):
    """Explicit restatement of F.interpolate(mode='bilinear', align_corners=False)."""
    B, C, H, W = x.shape

    def axis(n_in, n_out):
        dst = torch.arange(n_out, dtype=x.dtype)
        src = ((dst + 0.5) * (n_in / n_out) - 0.5).clamp(min=0)
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        t = src - i0.to(x.dtype)
        return i0, i1, t

    y0, y1, ty = axis(H, out_h)
    x0, x1, tx = axis(W, out_w)
    top = x[:, :, y0][:, :, :, x0] * (1 - tx) + x[:, :, y0][:, :, :, x1] * tx
    bot = x[:, :, y1][:, :, :, x0] * (1 - tx) + x[:, :, y1][:, :, :, x1] * tx
    return top * (1 - ty)[:, None] + bot * ty[:, None]


# --------------------------------------------------------------------------
# A8: coordinate Fourier input (lib/components/map3d_layers.py:260-275)


def synthesis_input(state, batch, height, width, dtype=torch.float32, prefix="synthesis_input"):
    ii = torch.linspace(-1, 1, height, dtype=dtype)[:, None].expand(height, width)
    jj = torch.linspace(-1, 1, width, dtype=dtype)[None, :].expand(height, width)
    coords = torch.stack([ii, jj], dim=0)[None].expand(batch, 2, height, width)
    w = state[f"{prefix}.network.0.weight"].to(dtype)
    b = state[f"{prefix}.network.0.bias"].to(dtype)
    return torch.sin(F.conv2d(coords, w, b))


# --------------------------------------------------------------------------
# A9: SPADE synthesis network (lib/generators/map3d_generator.py:58-97,
# lib/components/map3d_layers.py:176-190, 218-238, 346-352), eval mode.


def spectral_weight(state, prefix, training=False, buffers_out=None):
    """weight_orig / sigma.  eval: the *stored* u, v (no power iteration; torch.nn.utils.spectral_norm hook).
    training: one power iteration first (v <- normalize(W^T u), u <- normalize(W v), both without gradient, eps 1e-12),
    sigma = u . (W v) differentiable through W only; the new u, v are reported in buffers_out."""
    w = state[prefix + ".weight_orig"]
    u = state[prefix + ".weight_u"]
    v = state[prefix + ".weight_v"]
    wm = w.flatten(1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        if buffers_out is not None:
            buffers_out[prefix + ".weight_u"] = u
            buffers_out[prefix + ".weight_v"] = v
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def _bn_eval(state, prefix, x, eps=1e-5):
    dt = x.dtype
    mean = state[prefix + ".running_mean"].to(dt).view(1, -1, 1, 1)
    var = state[prefix + ".running_var"].to(dt).view(1, -1, 1, 1)
    g = state[prefix + ".weight"].to(dt).view(1, -1, 1, 1)
    b = state[prefix + ".bias"].to(dt).view(1, -1, 1, 1)
    return (x - mean) * torch.rsqrt(var + eps) * g + b


def _bn_train(state, prefix, x, buffers_out=None, eps=1e-5, momentum=0.1):
    """nn.SyncBatchNorm / BatchNorm2d in training mode (single process): batch mean and BIASED variance normalise,
    running statistics move by `momentum` toward the batch mean / UNBIASED variance, num_batches_tracked += 1."""
    dt = x.dtype
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
    if buffers_out is not None:
        with torch.no_grad():
            rm, rv = state[prefix + ".running_mean"].to(dt), state[prefix + ".running_var"].to(dt)
            buffers_out[prefix + ".running_mean"] = rm + momentum * (mean - rm)
            buffers_out[prefix + ".running_var"] = rv + momentum * (var * n / max(n - 1, 1) - rv)
            buffers_out[prefix + ".num_batches_tracked"] = state[prefix + ".num_batches_tracked"] + 1
    g = state[prefix + ".weight"].to(dt).view(1, -1, 1, 1)
    b = state[prefix + ".bias"].to(dt).view(1, -1, 1, 1)
    return (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + eps) * g + b


def _spade(state, prefix, x, style_map, training=False, buffers_out=None):
    dt = x.dtype
    if training:
        n = _bn_train(state, prefix + ".first_norm", x, buffers_out)
    else:
        n = _bn_eval(state, prefix + ".first_norm", x)
    a = torch.relu(F.conv2d(style_map, state[prefix + ".mlp_shared.0.weight"].to(dt),
                            state[prefix + ".mlp_shared.0.bias"].to(dt)))
    gamma = 1 + F.conv2d(a, state[prefix + ".mlp_gamma.weight"].to(dt), state[prefix + ".mlp_gamma.bias"].to(dt))
    beta = F.conv2d(a, state[prefix + ".mlp_beta.weight"].to(dt), state[prefix + ".mlp_beta.bias"].to(dt))
    return n * gamma + beta


def spade_block(state, prefix, x, style_map, skip, training=False, buffers_out=None):
    dt = x.dtype
    tb = (training, buffers_out)
    h = F.leaky_relu(_spade(state, prefix + ".spade_0", x, style_map, *tb), 0.2)
    h = F.conv2d(h, spectral_weight(state, prefix + ".conv_0", *tb).to(dt), state[prefix + ".conv_0.bias"].to(dt))
    h = F.leaky_relu(_spade(state, prefix + ".spade_1", h, style_map, *tb), 0.2)
    h = F.conv2d(h, spectral_weight(state, prefix + ".conv_1", *tb).to(dt), state[prefix + ".conv_1.bias"].to(dt))
    # reference compares the *last spatial dim* (width) of x and x_orig, which is always equal
    return h + x if skip else h


def synthesis_network(state, x, feature_maps, fixed_style, map3d_mode="mixed", mod_blocks=(0, 1, 2),
                      num_blocks=9, prefix="synthesis_network", return_internal=False, training=False, buffers_out=None):
    """x [B,C,H,W] (A8 output), feature_maps [B,F,H,W] (A7 output), fixed_style [B,1,F].
    training=True: train-mode semantics of the reference module (batch-statistics BatchNorm, one spectral-norm power
    iteration per conv and call); the buffers a train-mode forward would overwrite are returned through buffers_out."""
    B, _, H, W = x.shape
    fixed_map = fixed_style.reshape(B, -1, 1, 1).to(x.dtype).expand(B, fixed_style.shape[-1], H, W)
    rgb = None
    internals = {}
    for idx in range(num_blocks):
        if map3d_mode == "all":
            style_map = feature_maps + fixed_map
        elif map3d_mode == "mixed":
            style_map = feature_maps + fixed_map if idx in mod_blocks else fixed_map
        elif map3d_mode == "isolated":
            style_map = feature_maps if idx in mod_blocks else fixed_map
        else:
            raise ValueError("invalid map3d_mode")
        name = f"m3d_{idx}"
        x = spade_block(state, f"{prefix}.network.{name}", x, style_map, skip=idx >= num_blocks // 2, training=training,
                        buffers_out=buffers_out)
        if idx >= num_blocks // 2 - 1:
            dt = x.dtype
            o = F.conv2d(x, state[f"{prefix}.to_rgbs.{name}.linear.weight"].to(dt),
                         state[f"{prefix}.to_rgbs.{name}.linear.bias"].to(dt))
            rgb = o if rgb is None else o + rgb
        if return_internal:
            internals[name + "_feature_map"] = x
            internals[name + "_rgb"] = rgb
    internals["final"] = rgb
    return internals


# --------------------------------------------------------------------------
# A1-A11 assembled: Map3DGenerator.forward / staged_forward
# (lib/generators/map3d_generator.py:208-280, 282-378, 381-523)


def render(state, cfg, freq, phase, cond, jitter, noise, ray_subset=None):
    """ray_subset: LongTensor of ray indices -> the same computation for those rays only; the image-shaped outputs
    then come back as [B, C, 1, len(ray_subset)] (a one-row "image" over the chosen rays)."""
    hr, wr, S = cfg["render_height"], cfg["render_width"], cfg["num_steps"]
    focals = cond["intrinsics"][:, 0, 0]
    scales = cond["scales"].to(focals.dtype)
    pts, z_vals, dirs = ray_setup(focals, scales, cond["cam2world_matrices"], hr, wr, S,
                                  cfg["ray_start"], cfg["ray_end"], jitter,
                                  cfg.get("lock_view_dependence", False), ray_subset)
    if ray_subset is not None:
        if noise is not None:
            noise = noise[:, ray_subset]
        hr, wr = 1, int(ray_subset.numel())
    geo = geo_features(pts, cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"],
                       cond["fk_matrices"], cond["lbs_weights"], cfg.get("legacy_mode", False))
    field = neural_field(state, pts, freq, phase, geo, dirs, input_scaler=2.0 / cfg["side_length"],
                         num_blocks=cfg.get("neural_field_blocks", 4))
    B = pts.shape[0]
    field = field.reshape(B, hr * wr, S, -1)
    out, depth, w = ray_integration(field, z_vals, noise, cfg["clamp_mode"],
                                    cfg.get("last_back", False), cfg.get("white_back", False))
    img = out.reshape(B, hr, wr, -1).permute(0, 3, 1, 2)
    return img[:, :3] * 2 - 1, img[:, 3:], depth, w, dict(points=pts, z_vals=z_vals, geo=geo, field=field)


def sample_pdf(bins, weights, u, eps=1e-5):
    """Inverse-CDF sampling of lib/generators/volume_rendering.py:261-303 with the uniform draws `u` [N_rays, N_importance]
    passed in (the reference draws them with torch.rand at :285).  bins [N_rays, n+1], weights [N_rays, n]."""
    n = weights.shape[1]
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n)
    sel = torch.stack([below, above], -1).view(u.shape[0], -1)
    cdf_g = torch.gather(cdf, 1, sel).view(u.shape[0], -1, 2)
    bins_g = torch.gather(bins, 1, sel).view(u.shape[0], -1, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def ray_world_frame(focals, cam2world, render_h, render_w):
    """World-space ray origins [B,3] and unit directions [B,R,3] of the weak-perspective camera
    (volume_rendering.py:86-110, 133-170: transformed_ray_origins / transformed_ray_directions)."""
    B, dt, R = focals.shape[0], focals.dtype, render_h * render_w
    span = render_w / render_h
    px = torch.linspace(-span, span, render_w, dtype=dt).repeat(render_h)
    py = torch.linspace(-1, 1, render_h, dtype=dt).repeat_interleave(render_w)
    d = torch.stack([px.expand(B, R), py.expand(B, R), focals[:, None].expand(B, R)], dim=-1)
    d = d / (torch.norm(d, dim=-1, keepdim=True) + 1e-12)
    wd = torch.bmm(cam2world[:, :3, :3].to(dt), d.transpose(1, 2)).transpose(1, 2)
    return cam2world[:, :3, 3].to(dt), wd


def render_hierarchical(state, cfg, freq, phase, cond, jitter, noise_coarse, u, noise):
    """Map3DGenerator.render with hierarchical_sample=True (map3d_generator.py:449-516): coarse pass -> weights ->
    importance samples (sample_pdf) -> fine pass -> merge by depth -> integration over coarse + fine samples.
    noise_coarse [B,R,S,1] / noise [B,R,2S,1]: the (already scaled) integration noise of the two ray_integration
    calls, u [B*R, S]: the uniform draws of sample_pdf."""
    hr, wr, S = cfg["render_height"], cfg["render_width"], cfg["num_steps"]
    focals = cond["intrinsics"][:, 0, 0]
    scales = cond["scales"].to(focals.dtype)
    lock = cfg.get("lock_view_dependence", False)
    pts, z_vals, dirs = ray_setup(focals, scales, cond["cam2world_matrices"], hr, wr, S, cfg["ray_start"], cfg["ray_end"],
                                  jitter, lock)
    B, R = pts.shape[0], hr * wr
    legacy = cfg.get("legacy_mode", False)
    mesh = (cond["skeletons_xyz"], cond["vertices"], cond["tpose_vertices"], cond["fk_matrices"], cond["lbs_weights"])
    nb = cfg.get("neural_field_blocks", 4)
    scaler = 2.0 / cfg["side_length"]
    coarse = neural_field(state, pts, freq, phase, geo_features(pts, *mesh, legacy), dirs, input_scaler=scaler,
                          num_blocks=nb).reshape(B, R, S, -1)
    _, _, w = ray_integration(coarse, z_vals, noise_coarse, cfg["clamp_mode"], False, False)
    w = w.reshape(B * R, S) + 1e-5
    zv = z_vals.reshape(B * R, S)
    z_mid = 0.5 * (zv[:, :-1] + zv[:, 1:])
    fine_z = sample_pdf(z_mid, w[:, 1:-1], u).reshape(B, R, S, 1)
    origin, wd = ray_world_frame(focals, cond["cam2world_matrices"], hr, wr)
    fine_pts = (origin[:, None, None, :] + wd[:, :, None, :] * fine_z).reshape(B, R * S, 3)
    fine = neural_field(state, fine_pts, freq, phase, geo_features(fine_pts, *mesh, legacy), dirs, input_scaler=scaler,
                        num_blocks=nb).reshape(B, R, S, -1)
    all_out = torch.cat([fine, coarse], dim=-2)
    all_z = torch.cat([fine_z, z_vals], dim=-2)
    _, idx = torch.sort(all_z, dim=-2, stable=True)
    all_z = torch.gather(all_z, -2, idx)
    all_out = torch.gather(all_out, -2, idx.expand(-1, -1, -1, all_out.shape[-1]))
    out, depth, wts = ray_integration(all_out, all_z, noise, cfg["clamp_mode"], cfg.get("last_back", False),
                                      cfg.get("white_back", False))
    img = out.reshape(B, hr, wr, -1).permute(0, 3, 1, 2)
    return img[:, :3] * 2 - 1, img[:, 3:], depth, wts, dict(points=pts, z_vals=z_vals, fine_z=fine_z, all_z=all_z,
                                                          coarse_weights=w, field=all_out)


def generator_forward(state, cfg, z, cond, jitter, noise=None, truncation=None, return_internal=False, hier=None,
                      training=False, buffers_out=None, latent_indices=None):
    """truncation: None or (psi, avg_z, avg_freq, avg_phase, avg_styles) as produced by
    generate_avg_latent (map3d_generator.py:182-194, 295-301)."""
    if latent_indices is not None:                       # map3d_generator.py:218-219: latents come from the pool
        z = state["latent_pool.latents"][latent_indices]
    B = z.shape[0]
    zin = z if cfg.get("neural_field_latent_input", True) else torch.zeros_like(z)
    freq, phase = film_mapping(state, zin)
    _, styles = style_mapping(state, z)
    if truncation is not None:
        psi, _, af, ap, ast = truncation
        freq = af + psi * (freq - af)
        phase = ap + psi * (phase - ap)
        styles = ast + psi * (styles - ast)
    if hier is not None:          # hierarchical_sample=True: hier = dict(noise_coarse=..., u=...), noise covers 2S samples
        rgb_render, fmap, depth, w, inter = render_hierarchical(state, cfg, freq, phase, cond, jitter, hier["noise_coarse"],
                                                                hier["u"], noise)
    else:
        rgb_render, fmap, depth, w, inter = render(state, cfg, freq, phase, cond, jitter, noise)
    H, W = cfg["gen_height"], cfg["gen_width"]
    fmap_up = F.interpolate(fmap, (H, W), mode="bilinear")
    x0 = synthesis_input(state, B, H, W, dtype=z.dtype if z.dtype == torch.float64 else torch.float32)
    syn = synthesis_network(state, x0, fmap_up, styles, cfg.get("map3d_mode", "isolated"),
                            tuple(cfg["mod_blocks"]), cfg["synthesis_blocks"], return_internal=return_internal,
                            training=training, buffers_out=buffers_out)
    focals = cond["intrinsics"][:, 0, 0]
    zc = focals / cond["scales"].to(focals.dtype)
    dm = ((depth - zc.view(B, 1, 1)) / (cfg["depth_length"] / 2.0)).clamp(-1, 1)
    out = dict(rgbs=syn["final"], rgbs_render=rgb_render,
               depths=dm.reshape(B, cfg["render_height"], cfg["render_width"]).unsqueeze(1),
               freq=freq, phase=phase, styles=styles, feature_maps=fmap, feature_maps_up=fmap_up,
               weights=w, raw_depth=depth, x0=x0)
    out.update(inter)
    if return_internal:
        out.update({k: v for k, v in syn.items() if k != "final"})
    return out


def _resize_axis(n_in, n_out, dtype=torch.float32):
    """Source taps of F.interpolate(mode='bilinear', align_corners=False) along one axis (same formula as
    bilinear_resize above): i0, i1, t per destination index."""
    dst = torch.arange(n_out, dtype=dtype)
    src = ((dst + 0.5) * (n_in / n_out) - 0.5).clamp(min=0)
    i0 = src.floor().long().clamp(max=n_in - 1)
    i1 = (i0 + 1).clamp(max=n_in - 1)
    return i0, i1, src - i0.to(dtype)


def pixels_of_cells(cells, gen_hw, render_hw):
    """Output pixels whose upper-left bilinear tap is one of the low-resolution cells [(cy, cx), ...] -> sorted
    LongTensor of pixel indices p = Y*W + X.  Used to pick subsets that need only 4 rays per cell."""
    (H, W), (Hr, Wr) = gen_hw, render_hw
    y0, _, _ = _resize_axis(Hr, H)
    x0, _, _ = _resize_axis(Wr, W)
    pix = []
    for cy, cx in cells:
        ys = torch.nonzero(y0 == cy).flatten()
        xs = torch.nonzero(x0 == cx).flatten()
        if len(ys) and len(xs):
            pix.append((ys[:, None] * W + xs[None, :]).flatten())
    return torch.unique(torch.cat(pix)) if pix else torch.zeros(0, dtype=torch.long)


def generator_forward_subset(state, cfg, z, cond, jitter, pixel_subset, noise=None, truncation=None):
    """generator_forward restricted to the output pixels `pixel_subset` (LongTensor of p = Y*W + X, the same set for
    every batch item).  The path is per-ray up to the rendered feature maps and per-pixel after the bilinear resize, so
    this is the full computation on the rays / pixels the subset touches -- nothing is approximated; it exists so that
    BASELINE-size workloads (B=16, 512^2, 96x96 rays x 64 samples ...) can be checked in seconds.  jitter / noise are the
    FULL-size tensors.  -> dict(rgbs [B,3,P], rgbs_render [B,3,Rs], ray_subset [Rs] (sorted ray indices the pixels
    need), raw_depth [B,Rs,1], weights [B,Rs,S,1], feature_maps [B,F,Rs], sigma [B,Rs,S] (densities as integrated: noise
    added, before the clamp), taps [4,P] (positions in ray_subset of each pixel's four bilinear taps))."""
    B = z.shape[0]
    H, W = cfg["gen_height"], cfg["gen_width"]
    Hr, Wr = cfg["render_height"], cfg["render_width"]
    dt = z.dtype if z.dtype == torch.float64 else torch.float32
    Y, X = pixel_subset // W, pixel_subset % W
    y0, y1, ty = _resize_axis(Hr, H, dt)
    x0, x1, tx = _resize_axis(Wr, W, dt)
    taps = torch.stack([y0[Y] * Wr + x0[X], y0[Y] * Wr + x1[X], y1[Y] * Wr + x0[X], y1[Y] * Wr + x1[X]])   # [4,P]
    rays = torch.unique(taps)
    pos = torch.searchsorted(rays, taps)                                                                      # [4,P]
    zin = z if cfg.get("neural_field_latent_input", True) else torch.zeros_like(z)
    freq, phase = film_mapping(state, zin)
    _, styles = style_mapping(state, z)
    if truncation is not None:
        psi, _, af, ap, ast = truncation
        freq = af + psi * (freq - af)
        phase = ap + psi * (phase - ap)
        styles = ast + psi * (styles - ast)
    rgb_render, fmap, depth, w, internal = render(state, cfg, freq, phase, cond, jitter, noise, ray_subset=rays)
    sigma = internal["field"][..., -1]                                                                        # [B,Rs,S]
    if noise is not None:
        sigma = sigma + noise[:, rays, :, 0]
    fm = fmap[:, :, 0, :]                                                                                     # [B,F,Rs]
    txp, typ = tx[X], ty[Y]
    top = fm[:, :, pos[0]] * (1 - txp) + fm[:, :, pos[1]] * txp
    bot = fm[:, :, pos[2]] * (1 - txp) + fm[:, :, pos[3]] * txp
    fmap_up = (top * (1 - typ) + bot * typ).unsqueeze(-1)                                                     # [B,F,P,1]
    # A8 at the chosen pixels (synthesis_input restricted to rows of the coordinate grid)
    ii = torch.linspace(-1, 1, H, dtype=dt)[Y]
    jj = torch.linspace(-1, 1, W, dtype=dt)[X]
    coords = torch.stack([ii, jj], dim=0)[None, :, :, None].expand(B, 2, len(pixel_subset), 1)
    x0in = torch.sin(F.conv2d(coords, state["synthesis_input.network.0.weight"].to(dt),
                              state["synthesis_input.network.0.bias"].to(dt)))
    syn = synthesis_network(state, x0in, fmap_up, styles, cfg.get("map3d_mode", "isolated"),
                            tuple(cfg["mod_blocks"]), cfg["synthesis_blocks"])
    return dict(rgbs=syn["final"][..., 0], rgbs_render=rgb_render[:, :, 0, :], ray_subset=rays, raw_depth=depth,
                weights=w, feature_maps=fm, styles=styles, freq=freq, phase=phase, sigma=sigma, taps=pos)


# --------------------------------------------------------------------------
# P2: upfirdn2d reference semantics (lib/components/ops/upfirdn2d.py:166-210)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Direct tap-loop definition: zero-stuff by `up`, pad/crop, correlate with the
    (flipped unless flip_filter) filter, keep every `down`-th sample.
    out size = (in*up + pad0 + pad1 - taps + down) // down  (upfirdn2d.cpp:35-36)."""
    upx, upy = (up, up) if isinstance(up, int) else up
    downx, downy = (down, down) if isinstance(down, int) else down
    if isinstance(padding, int):
        padding = [padding] * 4
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    px0, px1, py0, py1 = padding
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    B, C, H, W = x.shape
    k2 = (f[:, None] * f[None, :]) if f.ndim == 1 else f
    if f.ndim == 1:
        k2 = k2 * gain          # gain**(1/2) per 1-D pass, two passes
    else:
        k2 = k2 * gain
    k2 = k2.to(x.dtype)
    if not flip_filter:
        k2 = k2.flip([0, 1])
    fh, fw = k2.shape
    UH, UW = H * upy + py0 + py1, W * upx + px0 + px1
    canvas = x.new_zeros(B, C, H * upy + max(py0, 0) + max(py1, 0), W * upx + max(px0, 0) + max(px1, 0))
    canvas[:, :, max(py0, 0):max(py0, 0) + H * upy:upy, max(px0, 0):max(px0, 0) + W * upx:upx] = x
    canvas = canvas[:, :, max(-py0, 0): canvas.shape[2] - max(-py1, 0), max(-px0, 0): canvas.shape[3] - max(-px1, 0)]
    assert canvas.shape[2] == UH and canvas.shape[3] == UW
    FH, FW = UH - fh + 1, UW - fw + 1
    full = x.new_zeros(B, C, FH, FW)
    for a in range(fh):
        for b in range(fw):
            full = full + k2[a, b] * canvas[:, :, a:a + FH, b:b + FW]
    return full[:, :, ::downy, ::downx]


def setup_filter(taps, normalize=True, flip_filter=False, gain=1, separable=None):
    """lib/components/ops/upfirdn2d.py:69-113."""
    if taps is None:
        taps = 1
    f = torch.as_tensor(taps, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


# --------------------------------------------------------------------------
# P3: modulated convs
# per-pixel modulated 1x1 (lib/components/map3d_layers.py:60-80)


def modconv1x1_pixelwise(x, style, weight, bias, affine_w, affine_b, demodulate=True, eps=1e-8):
    """x [B,P,Cin], style [B,P,S], weight [Cin,Cout], bias [Cout]."""
    m = F.linear(style, affine_w, affine_b) + 1
    w = weight[None, None] * m.unsqueeze(-1)
    if demodulate:
        w = w * torch.rsqrt(w.pow(2).sum(dim=2, keepdim=True) + eps)
    return (x.unsqueeze(-1) * w).sum(dim=2) + bias


# StyleGAN2 grouped modulated conv (lib/components/cips_layers.py:235-278)


def modconv2d_grouped(x, style, weight, bias, mod_w, mod_b, demodulate=True, eps=1e-8):
    """x [B,Cin,H,W], style [B,S], weight [1,Cout,Cin,k,k], bias [1,Cout]; per-sample
    weights w_b = weight * (affine(style_b) + 1), optionally demodulated over (Cin,k,k)."""
    B = x.shape[0]
    k = weight.shape[-1]
    outs = []
    for b in range(B):
        s = F.linear(style[b:b + 1], mod_w, mod_b).view(1, -1, 1, 1) + 1.0
        w = weight[0] * s
        if demodulate:
            w = w * torch.rsqrt(w.pow(2).sum([1, 2, 3], keepdim=True) + eps)
        outs.append(F.conv2d(x[b:b + 1], w, padding=k // 2))
    return torch.cat(outs, 0) + bias.view(1, -1, 1, 1)
