"""The differentiable (training) evaluation of the generator path -- SURVEY 8f.4, "backward passes of A4-A9".

The inference engines (csrc/field_x3*.hip, synthesis_x3*.hip) keep every activation in registers / LDS and therefore have
nothing to differentiate through; training needs the activations in HBM anyway.  This path is organised for that:

  * forward and data-gradient contractions are library GEMMs over ALL samples / pixels of the batch at once
    (``[B*N, C] x [C, C']``, hipBLASLt through torch: 0.5 M x 256 x 256 problems -- the shape rocBLAS is built for), in a
    channels-LAST layout end to end, so no transposes sit between the render and the synthesis network; the WEIGHT gradients
    (contraction over the 0.5 M rows, tiny output) run on the hand-written split-K matrix-core kernel of csrc/wgrad_x3.hip
    (``ops.linear``);
  * what sits between two GEMMs is ONE hand-written HIP pass each way with a hand-written adjoint that recomputes instead of
    storing: ``film_sin`` (sine activation with per-sample frequency / phase), ``h3d_ray_integrate`` /
    ``h3d_ray_integrate_bwd`` (volume integration), ``bias_act`` (style mapping network), and the SPADE normalise-modulate-
    activate chain (``spade_norm_act``: one pass forward, two backward, csrc/spade_train.hip);
  * ray set-up and the SMPL geometry features have no learnable inputs and run as the same HIP kernels as in inference.

Train-mode semantics of the reference are reproduced: batch-statistics BatchNorm (synchronised over the process group when
one is initialised -- the reference uses nn.SyncBatchNorm, lib/components/map3d_layers.py:162) with running-statistics
updates, and one spectral-norm power iteration per conv and call (torch.nn.utils.spectral_norm as used at
lib/components/map3d_layers.py:205-206).
"""
import os

import torch
import torch.nn.functional as F

from ..components.ops.film import film_sin
from ..components.ops.linear import linear
from ..components.ops.spade import spade_norm_act
from ..components.ops.spectral import _SpectralWeight
from ..components.resample import bilinear_resize_cl, bilinear_resize_relu_cl

FUSED_SPECTRAL = os.environ.get("H3D_GEN_SN", "hip") != "torch"      # the generator's spectral norm on csrc/spectral_norm.hip (round 6)
ALIAS_GRADS = os.environ.get("H3D_SPADE_ALIAS", "1") != "0"      # gradients of a skip block's input summed inside the SPADE backward kernel (round 6)


# ------------------------------------------------------------------------------------------------ A5: the implicit function

def field_forward(nf, points, freq, phase, geo, dirs, input_scaler=1.0, geo_feature_scaler=1.0):
    """COORDCONCATSIREN.forward (lib/implicit_funcitions/modulated.py:41-75), differentiable w.r.t. the module's parameters,
    freq and phase.  points [B,N,3], geo [B,N,31], dirs [B,N,3] or None (= the locked direction (0,0,-1)),
    freq / phase [B,4H] -> [B,N,F+4] = [rgb, features, sigma]."""
    H = nf.hidden_dim
    fr = freq * 15 + 30
    a = film_sin(linear(points * input_scaler, nf.first_layer_coord.layer.weight, nf.first_layer_coord.layer.bias), w0=30.0)
    g = film_sin(linear(geo if geo_feature_scaler == 1.0 else geo * geo_feature_scaler, nf.first_layer_mod.layer.weight,
                          nf.first_layer_mod.layer.bias), w0=30.0)
    x = torch.cat([a, g], dim=-1)
    for k, dense in enumerate(nf.network):
        sl = slice(k * H, (k + 1) * H)
        x = film_sin(linear(x, dense.layer.weight, dense.layer.bias), fr[:, sl], phase[:, sl])
    sigma = linear(x, nf.sigma_layer.weight, nf.sigma_layer.bias)
    wc, bc = nf.color_layer_sine.layer.weight, nf.color_layer_sine.layer.bias
    if dirs is None:                       # lock_view_dependence: the direction is the constant (0,0,-1) -> part of the bias
        c = linear(x, wc[:, 3:], bc - wc[:, 2])
    else:
        c = linear(x, wc[:, 3:], bc, add=linear(dirs, wc[:, :3]))       # the view-direction term joins in the GEMM's epilogue
    c = film_sin(c, fr[:, -H:], phase[:, -H:])
    rgb = torch.sigmoid(linear(c, nf.color_layer_linear.weight, nf.color_layer_linear.bias))
    feat = linear(c, nf.feature_layer_linear.weight, nf.feature_layer_linear.bias)
    return torch.cat([rgb, feat, sigma], dim=-1)


# ------------------------------------------------------------------------------------------------ spectral norm

def spectral_weight(conv, training, eps=1e-12):
    """weight_orig / sigma of a spectral-normalised 1x1 conv holder (bias, weight_orig, weight_u, weight_v) as a [Cout, Cin]
    matrix.  training: one power iteration first, the new u / v overwrite the buffers (torch.nn.utils.spectral_norm)."""
    with torch.autocast("cuda", enabled=False):          # under AMP the power iteration and sigma stay in fp32
        return _spectral_weight_fp32(conv, training, eps)


def _spectral_weight_fp32(conv, training, eps):
    w0, u, v = conv.weight_orig, conv.weight_u, conv.weight_v
    if (training and FUSED_SPECTRAL and w0.is_cuda and w0.dtype == torch.float32 and u.dtype == torch.float32 and v.dtype == torch.float32
            and u.is_contiguous() and v.is_contiguous()):
        # round 6: the discriminator's fused kernels (ops/spectral.py: three launches forward, two backward) instead of ~16 + ~10
        # tensor operations per layer and pass -- 36 calls per config-4 iteration
        return _SpectralWeight.apply(w0, u, v, eps).flatten(1)
    w = conv.weight_orig.flatten(1).float()
    if training:
        with torch.no_grad():
            v_new = F.normalize(torch.mv(w.t(), u), dim=0, eps=eps)
            u_new = F.normalize(torch.mv(w, v_new), dim=0, eps=eps)
            conv.weight_v.copy_(v_new)
            conv.weight_u.copy_(u_new)
        u, v = u_new, v_new
    sigma = torch.dot(u, torch.mv(w, v))
    return w / sigma


# ------------------------------------------------------------------------------------------------ A7-A9: synthesis

def _coords(H, W, device, dtype):
    ii = torch.linspace(-1, 1, H, device=device, dtype=dtype)[:, None].expand(H, W)
    jj = torch.linspace(-1, 1, W, device=device, dtype=dtype)[None, :].expand(H, W)
    return torch.stack([ii, jj], dim=-1).reshape(H * W, 2)


FUSED_RESIZE_RELU = os.environ.get("H3D_RESIZE_RELU", "fused") != "torch"      # ReLU + its mask inside the resize kernels (round 6; A/B switch)


def _resize_channels_last(t, render_hw, gen_hw, relu=False):
    """Bilinear (align_corners=False) resize of a channels-last map [B, Hr*Wr, C] -> [B, H*W, C] without leaving the
    channels-last layout (F.interpolate on the NCHW *view* of the same memory)."""
    B, _, C = t.shape
    if t.is_cuda and t.dtype in (torch.float32, torch.float16) and C % 4 == 0 and B * gen_hw[0] < 65536:
        # own kernels: the backward reads the gradient once, no atomics.  Under autocast too (F.interpolate's channels-last
        # kernels take 3.4 ms forward + 4.9 ms backward here, these 0.5 + 0.3 ms): computed in fp32, returned in the input's type
        if relu and FUSED_RESIZE_RELU and t.dtype == torch.float32:
            # fp32 only: under float16 autocast the mask would be the saved fp32 output (twice the bytes of the f16 ReLU's) and
            # the iteration gets 1.5 ms slower (profiles/r6_ab_resize_relu.txt)
            return bilinear_resize_relu_cl(t, render_hw, gen_hw)
        up = bilinear_resize_cl(t.float(), render_hw, gen_hw).to(t.dtype)
        return torch.relu(up) if relu else up
    nchw = t.reshape(B, render_hw[0], render_hw[1], C).permute(0, 3, 1, 2)
    up = F.interpolate(nchw, gen_hw, mode="bilinear", align_corners=False)
    up = up.permute(0, 2, 3, 1).reshape(B, gen_hw[0] * gen_hw[1], C)
    return torch.relu(up) if relu else up


def synthesis_forward(G, fmap_low, styles, render_hw, gen_hw, training, group=None, spade_kernels=None):
    """SynthesisInput + bilinear resize + SynthesisNetwork (lib/generators/map3d_generator.py:58-97, 244-275;
    lib/components/map3d_layers.py:176-275, 346-352).  fmap_low [B,R,F] channels-last rendered features, styles [B,1,F]
    -> rgb [B,3,H,W]."""
    sn = G.synthesis_network
    B, _, Fd = fmap_low.shape
    H, W = gen_hw
    P = H * W
    dev, dt = fmap_low.device, fmap_low.dtype
    fixed = styles.reshape(B, 1, Fd)
    mode, mod_blocks, nb = sn.map3d_mode, set(sn.mod_blocks), sn.num_blocks
    if mode not in ("all", "mixed", "isolated"):
        raise ValueError("invalid map3d_mode")
    conv_in = G.synthesis_input.network[0]
    x0 = torch.sin(linear(_coords(H, W, dev, dt), conv_in.weight.flatten(1), conv_in.bias))      # [P, F]
    x = x0.unsqueeze(0).expand(B, P, x0.shape[-1])

    def per_pixel(idx):
        return mode == "all" or idx in mod_blocks

    # the 128-wide shared convs of every per-pixel SPADE in ONE low-resolution GEMM + ONE resize (conv1x1 and the bilinear
    # resize commute).  The constant style's contribution -- a per-sample bias -- is added BEFORE the resize (round 6): the
    # resize's weights sum to one, so resize(v + c) = resize(v) + c, and the addition and its gradient's sum over the pixels run
    # over the 96 x 48 rays instead of the 512 x 256 pixels (2 ms per config-4 iteration).
    names = [f"m3d_{i}" for i in range(nb)]
    pix = [(n, s) for i, n in enumerate(names) if per_pixel(i) for s in ("spade_0", "spade_1")]
    shared_up = {}
    if pix:
        shared = [getattr(sn.network[n], s).mlp_shared[0] for n, s in pix]
        w_all = torch.cat([m.weight.flatten(1) for m in shared], dim=0)
        if mode == "isolated":             # the style is the feature map alone: the offsets are the biases
            low = linear(fmap_low, w_all, torch.cat([m.bias for m in shared], dim=0))
        else:
            off_all = torch.cat([linear(fixed, m.weight.flatten(1), m.bias) for m in shared], dim=-1)      # [B, 1, 128 * len(pix)]
            low = linear(fmap_low, w_all)
            low = low + off_all.to(low.dtype)
        # ... and so is the ReLU: one pass over the whole map instead of one per (strided) piece
        up = _resize_channels_last(low, render_hw, gen_hw, relu=True)                           # [B, P, 128 * len(pix)]; ReLU in the resize kernel
        # split, not slices: the backward of a split is ONE concatenation of the pieces' gradients; slices would each
        # zero-fill a full-width gradient and add them up (6 x 1.6 GB at config 4)
        shared_up = dict(zip(pix, torch.split(up, 128, dim=-1)))

    def modulation(blk_name, spade_name, idx):
        sp = getattr(sn.network[blk_name], spade_name)
        ws, bs = sp.mlp_shared[0].weight.flatten(1), sp.mlp_shared[0].bias
        if per_pixel(idx):
            a = shared_up[(blk_name, spade_name)]                              # offset and ReLU happened in front of the split
        else:
            a = torch.relu(linear(fixed, ws, bs))                              # [B,1,128]
        gamma = linear(a, sp.mlp_gamma.weight.flatten(1), sp.mlp_gamma.bias)
        beta = linear(a, sp.mlp_beta.weight.flatten(1), sp.mlp_beta.bias)
        return sp.first_norm, gamma, beta

    rgb = None
    pending = None      # ToRGB head of the previous block's output, evaluated on a view the next block's first SPADE hands out
    # Training mode: the batch moments every SPADE needs of its input come out of the epilogue of the GEMM that produced it
    # (h3d_conv_x3_moments, round 6) -- a [rows, 2, C] buffer of per-workgroup sums instead of a pass over [B, P, C] per SPADE.
    mom_x = None
    for idx, name in enumerate(names):
        blk = sn.network[name]
        skip = idx >= nb // 2
        # A skip block's input feeds its first SPADE, its residual connection and the previous block's ToRGB head.  When autograd
        # records, the last two read x through views handed out by the SPADE node, whose backward kernel adds their gradients into
        # dx as it writes it (h3d_spade_bwd_apply_acc) -- two accumulation passes over [B, P, C] less per block (round 6).
        n_alias = (int(skip) + int(pending is not None)) if (ALIAS_GRADS and torch.is_grad_enabled() and x.requires_grad) else 0
        out = spade_norm_act(x, *modulation(name, "spade_0", idx), training, group, kernels=spade_kernels, aliases=n_alias,
                             moments=mom_x)
        h, views = (out[0], list(out[1:])) if n_alias else (out, [])
        x_in = (views.pop(0) if views else x) if skip else None
        if pending is not None:
            o = linear(views.pop(0) if views else x, pending.weight.flatten(1), pending.bias)
            rgb = o if rgb is None else o + rgb
            pending = None
        h, mom_h = linear(h, spectral_weight(blk.conv_0, training), blk.conv_0.bias, moments=True) if training else \
            (linear(h, spectral_weight(blk.conv_0, training), blk.conv_0.bias), None)
        h = spade_norm_act(h, *modulation(name, "spade_1", idx), training, group, kernels=spade_kernels, moments=mom_h)
        # skip blocks: the residual connection joins in the GEMM's epilogue (h3d_conv_x3_add) instead of a pass of its own
        if training and idx + 1 < nb:
            x, mom_x = linear(h, spectral_weight(blk.conv_1, training), blk.conv_1.bias, add=x_in, moments=True)
        else:
            x, mom_x = linear(h, spectral_weight(blk.conv_1, training), blk.conv_1.bias, add=x_in), None
        if idx >= nb // 2 - 1:
            lin = sn.to_rgbs[name].linear
            if idx + 1 < nb:
                pending = lin
            else:
                o = linear(x, lin.weight.flatten(1), lin.bias)
                rgb = o if rgb is None else o + rgb
    return rgb.reshape(B, H, W, 3).permute(0, 3, 1, 2)
