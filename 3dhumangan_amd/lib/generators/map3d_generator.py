"""Map3DGenerator -- drop-in for the reference class of the same name
(lib/generators/map3d_generator.py:101-523): same constructor kwargs (the whole config dict is splatted in), same
state_dict key schema, same ``forward`` / ``staged_forward`` / ``set_device`` / ``generate_avg_latent`` surface and
output dict keys.  In `.eval()` mode every hot stage is a fused HIP kernel behind libh3d.so (in `.train()` mode the
differentiable path of lib/generators/differentiable.py runs instead, see Map3DGenerator.wants_autograd):

    mapping networks (B x L GEMVs)      torch GEMMs on the device + h3d_bias_act        (A1, A2)
    ray set-up                          h3d_ray_setup                                     (A3)
    SMPL geometry features              h3d_geo_features                                  (A4)
    FiLM-SIREN + volume integration     h3d_render_fused  (or h3d_neural_field + h3d_ray_integrate)   (A5, A6)
    resize + synthesis input + 9 SPADE blocks + ToRGB      h3d_synthesis                  (A7, A8, A9)

The reference forward is stochastic.  The draws happen here at the same places with torch's device
RNG; ``jitter=`` / ``noise=`` kwargs inject explicit tensors instead (used by the parity tests).
"""
import math
import os

import torch
import torch.nn as nn

from ... import _lib
from ..._stages import stage
from ..components import smpl
from ..components.ops.bias_act import bias_act
from . import volume_rendering as vr
from .differentiable import field_forward, synthesis_forward
from .synthesis_pack import SynthesisPlan


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def _consume_camera_rng(sample_dist, batch, device):
    """The reference samples a camera position inside transform_sampled_points and throws it away (the camera comes
    from the conditions): volume_rendering.py:141-143 -> sample_camera_positions (:182-221).  Only the RNG draws
    survive, and they depend on ``sample_dist``; they are mirrored here so that the integration noise drawn afterwards
    comes from the same place of the device RNG stream for every mode:
      'uniform', 'spherical_uniform'   two rand  [B,1]
      'normal', 'gaussian'             two randn [B,1]
      'truncated_gaussian'             two normal_ [B,1,4]   (truncated_normal_, :173-180)
      'hybrid'                         python's random.random() picks the rand or the randn pair
      anything else (None)             no draw"""
    if sample_dist in ("uniform", "spherical_uniform"):
        torch.rand((batch, 1), device=device), torch.rand((batch, 1), device=device)
    elif sample_dist in ("normal", "gaussian"):
        torch.randn((batch, 1), device=device), torch.randn((batch, 1), device=device)
    elif sample_dist == "hybrid":
        import random
        draw = torch.rand if random.random() < 0.5 else torch.randn
        draw((batch, 1), device=device), draw((batch, 1), device=device)
    elif sample_dist == "truncated_gaussian":
        torch.empty((batch, 1, 4), device=device).normal_(), torch.empty((batch, 1, 4), device=device).normal_()


class LatentPool(nn.Module):
    """reference lib/components/util.py:16-29."""

    def __init__(self, pool_size, latent_dim):
        super().__init__()
        self.latents = nn.Parameter(torch.zeros([pool_size, latent_dim]), requires_grad=True)

    def init(self, latents):
        with torch.no_grad():
            self.latents.copy_(latents)

    def forward(self, indices):
        return self.latents[indices]


class MappingNetwork(nn.Module):
    """FiLM frequency/phase mapping (reference lib/components/mapping_networks.py:13-41)."""

    def __init__(self, latent_dim, map_hidden_dim, map_output_dim):
        super().__init__()
        self.network = nn.Sequential(nn.Linear(latent_dim, map_hidden_dim), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(map_hidden_dim, map_hidden_dim), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(map_hidden_dim, map_hidden_dim), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(map_hidden_dim, map_output_dim))
        for m in self.network:
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        with torch.no_grad():
            self.network[-1].weight *= 0.25

    def forward(self, z):
        out = self.network(normalize_2nd_moment(z.to(torch.float32)))
        half = out.shape[-1] // 2
        return out[..., :half], out[..., half:]


class FullyConnectedLayer(nn.Module):
    """Equalised-learning-rate dense layer holding the parameters of the reference's FullyConnectedLayer
    (mapping_networks.py:92-121: weight stored divided by lr_multiplier, raw bias; same state_dict keys).  Inference
    form: the runtime gains are folded into an effective (W^T, b) pair once per weight version, the product is one
    library GEMM and bias + activation go through the HIP bias_act op."""

    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * (1.0 / lr_multiplier))
        self.bias = nn.Parameter(torch.full((out_features,), float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)      # callers may rescale it (implicit branch: x0.2)
        self.bias_gain = lr_multiplier
        self._folded = (None, None, None)

    def folded(self):
        """(W^T * weight_gain [in, out], b * bias_gain or None), rebuilt when a parameter or a gain changes."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight_gain, self.bias_gain,
               None if self.bias is None else (self.bias.data_ptr(), self.bias._version))
        if self._folded[0] != key:
            wt = (self.weight.detach().float() * self.weight_gain).t().contiguous()
            b = None if self.bias is None else self.bias.detach().float() * self.bias_gain
            self._folded = (key, wt, b)
        return self._folded[1], self._folded[2]

    def forward(self, x):
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            # training form: the gains stay in the graph (equalised learning rate), bias_act is differentiable
            wt = (self.weight.float() * self.weight_gain).t()
            b = None if self.bias is None else self.bias.float() * self.bias_gain
        else:
            wt, b = self.folded()
        # [B, 512] x [512, 512] products: under float16 autocast they stay in fp32 (round 6) -- autocast would convert the activation
        # and both parameters of every layer and call (the folded weight is a non-leaf, so its conversion is never cached) for a
        # product that costs nothing: 3 launches + 3 in the backward per layer, ~100 launches per iteration
        with torch.autocast("cuda", enabled=False):
            y = x.float() @ wt
            if self.activation == "linear":
                return y if b is None else y + b
            return bias_act(y, b, act=self.activation)


class TwoPartMappingNetwork(nn.Module):
    """reference mapping_networks.py:124-216 with c_dim == 0 (the only use)."""

    def __init__(self, z_dim, c_dim, implicit_dim, w_dim, num_ws, trunk_layers=6, branch_layers=2, activation="lrelu",
                 lr_multiplier=0.01, **_):
        super().__init__()
        if c_dim != 0:
            raise NotImplementedError("label conditioning (c_dim > 0) is not used by any config")
        self.z_dim, self.w_dim, self.num_ws = z_dim, w_dim, num_ws
        self.trunk_layers, self.branch_layers = trunk_layers, branch_layers
        dims = [z_dim] + [w_dim] * trunk_layers
        for i in range(trunk_layers):
            setattr(self, f"trunk{i}", FullyConnectedLayer(dims[i], dims[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        ich = [w_dim] * branch_layers + [implicit_dim]
        sch = [w_dim] * branch_layers + [w_dim]
        for i in range(branch_layers):
            setattr(self, f"implicit{i}", FullyConnectedLayer(ich[i], ich[i + 1], lr_multiplier=lr_multiplier,
                                                             activation="linear" if i == branch_layers - 1 else activation))
            setattr(self, f"superres{i}", FullyConnectedLayer(sch[i], sch[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        getattr(self, f"implicit{branch_layers - 1}").weight_gain *= 0.2

    def forward(self, z, c=None, **_):
        assert z.shape[1] == self.z_dim
        x = normalize_2nd_moment(z.to(torch.float32))
        for i in range(self.trunk_layers):
            x = getattr(self, f"trunk{i}")(x)
        xi, xs = x, x
        for i in range(self.branch_layers):
            xi = getattr(self, f"implicit{i}")(xi)
            xs = getattr(self, f"superres{i}")(xs)
        if self.num_ws is not None:
            xs = xs.unsqueeze(1).repeat([1, self.num_ws, 1])
        return xi, xs


# ---- parameter holders that reproduce the reference's state_dict paths for the synthesis side ---------------

class _SpectralConv(nn.Module):
    """nn.utils.spectral_norm(nn.Conv2d(cin, cout, 1)) as stored: bias, weight_orig, weight_u, weight_v."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cout, cin, 1, 1)
        nn.init.kaiming_normal_(w, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_orig = nn.Parameter(w)
        u, s, vh = torch.linalg.svd(w.flatten(1), full_matrices=False)
        self.register_buffer("weight_u", u[:, 0].clone())
        self.register_buffer("weight_v", vh[0].clone())


class _BatchNormStats(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


def _conv1x1(cin, cout):
    m = nn.Conv2d(cin, cout, kernel_size=1)
    nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
    return m


class _Spade(nn.Module):
    def __init__(self, c, style_dim):
        super().__init__()
        self.first_norm = _BatchNormStats(c)
        self.mlp_shared = nn.Sequential(_conv1x1(style_dim, 128), nn.ReLU())
        self.mlp_gamma = _conv1x1(128, c)
        self.mlp_beta = _conv1x1(128, c)


class _SpadeBlock(nn.Module):
    def __init__(self, cin, cout, style_dim):
        super().__init__()
        self.conv_0 = _SpectralConv(cin, cout)
        self.conv_1 = _SpectralConv(cout, cout)
        self.spade_0 = _Spade(cin, style_dim)
        self.spade_1 = _Spade(cout, style_dim)


class _ToRGB(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.linear = nn.Conv2d(c, 3, 1)
        with torch.no_grad():
            self.linear.weight *= 0.25


class SynthesisNetwork(nn.Module):
    """Parameters of reference map3d_generator.py:14-55; evaluation goes through SynthesisPlan / h3d_synthesis."""

    def __init__(self, input_dim, style_dim, hidden_dim=256, num_blocks=8, mod_blocks=list(range(8)), name_prefix="m3d",
                 spatial_normalization="instance_norm", map3d_mode="isolated", **_):
        super().__init__()
        if spatial_normalization != "batch_norm":
            raise NotImplementedError("only spatial_normalization='batch_norm' (all shipped configs) has a HIP kernel")
        self.style_dim, self.num_blocks, self.mod_blocks, self.map3d_mode = style_dim, num_blocks, list(mod_blocks), map3d_mode
        net, rgbs = {}, {}
        cin = input_dim
        for i in range(num_blocks):
            net[f"{name_prefix}_{i}"] = _SpadeBlock(cin, hidden_dim, style_dim)
            rgbs[f"{name_prefix}_{i}"] = _ToRGB(hidden_dim)
            cin = hidden_dim
        self.network = nn.ModuleDict(net)
        self.to_rgbs = nn.ModuleDict(rgbs)


class _CoordInput(nn.Module):
    """SynthesisInput parameters (reference map3d_layers.py:241-258): network.0 = Conv2d(2, F, 1)."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        conv = nn.Conv2d(input_dim, output_dim, kernel_size=1)
        nn.init.uniform_(conv.weight, -math.sqrt(9 / input_dim), math.sqrt(9 / input_dim))
        self.network = nn.Sequential(conv)


class _StyleInput(nn.Module):
    """SynthesisStyleInput parameters (reference map3d_layers.py:278-343); only reachable through
    disable_render=True, which no config sets -- kept so reference checkpoints load with strict=True."""

    def __init__(self, input_dim, latent_dim, output_dim):
        super().__init__()
        self.from_coords = nn.Sequential(nn.Conv2d(input_dim, latent_dim, 1))
        self.network = nn.Sequential(nn.Conv2d(latent_dim * 2, output_dim, 1), nn.LeakyReLU(0.2),
                                     nn.Conv2d(output_dim, output_dim, 1), nn.LeakyReLU(0.2))


class Map3DGenerator(nn.Module):

    def __init__(self, neural_field_cls, **kwargs):
        super().__init__()
        k = kwargs
        self.latent_dim, self.hidden_dim, self.feature_dim = k["latent_dim"], k["hidden_dim"], k["feature_dim"]
        self.geo_feature_dim, self.label_dim = k["geo_feature_dim"], k["label_dim"]
        self.gen_height, self.gen_width = k["gen_height"], k["gen_width"]
        self.disable_modulation = k.get("disable_modulation", False)
        self.legacy_mode = k.get("legacy_mode", False)
        # eval-mode render: nearest-vertex search + fused field kernel that builds the geometry features itself (default), or
        # h3d_geo_features + fused field kernel (H3D_FUSE_GEO=0 / .fuse_geo = False)
        self.fuse_geo = os.environ.get("H3D_FUSE_GEO", "1") != "0"
        # Train-mode forward that nothing records (the D step's no-grad generator forward, reference
        # lib/trainers/phase_trainer.py:355-362): the field + integration run on the fused render with device-packed weights (round 6).
        # "x3" (default): three f16 products, the differentiable path's own error class (~1e-5); "x2": the inference default's tier;
        # "off": lib/generators/differentiable.py as in rounds 2-5.  The field has no batch statistics: train and eval mode coincide.
        self.train_field = os.environ.get("H3D_TRAIN_FIELD", "x3")
        for flag in ("2d_semantic_input", "2d_label_input", "2d_latent_input"):
            if k.get(flag, False):
                raise NotImplementedError(f"{flag}=True is not used by any shipped config and has no HIP path")
        self.neural_field = neural_field_cls(output_dim=k["feature_dim"] + 4, latent_dim=k["latent_dim"],
                                             input_dim=k["input_dim"], hidden_dim=k["hidden_dim"],
                                             geo_feature_dim=k["geo_feature_dim"], feature_dim=k["feature_dim"],
                                             num_blocks=k["neural_field_blocks"], device=None)
        self.synthesis_input = _CoordInput(2, k["feature_dim"])
        self.synthesis_style_input = _StyleInput(1 if "segments" in k["condition_modal_gen"] else 3, k["latent_dim"],
                                                 k["feature_dim"])
        self.synthesis_network = SynthesisNetwork(input_dim=k["feature_dim"], style_dim=k["feature_dim"],
                                                  hidden_dim=k["hidden_dim"], num_blocks=k["synthesis_blocks"],
                                                  mod_blocks=k["mod_blocks"], map3d_mode=k.get("map3d_mode", "isolated"),
                                                  spatial_normalization=k.get("spatial_normalization", "instance_norm"))
        self.neural_field_mapping_network = MappingNetwork(latent_dim=k["latent_dim"], map_hidden_dim=k["hidden_dim"],
                                                           map_output_dim=2 * k["neural_field_blocks"] * k["hidden_dim"])
        self.synthesis_mapping_network = TwoPartMappingNetwork(z_dim=k["latent_dim"], c_dim=0, implicit_dim=1,
                                                               w_dim=k["feature_dim"], num_ws=1, trunk_layers=7,
                                                               branch_layers=1, lr_multiplier=0.01)
        self.epoch = 0
        self.step = 0
        self.side_length = k["side_length"]
        self.latent_pool = LatentPool(k["dataset_length"], k["latent_dim"])
        self.device = None
        self.avg_latent = None
        self._avg_latent_key = None
        self._plan = None
        self._plan_key = None
        self.stage_timer = None          # set to _stages.StageTimer() to collect per-stage HIP-event timings

    # ------------------------------------------------------------------ reference surface
    def set_device(self, device):
        self.device = device
        self.neural_field.device = device

    def _mapping_key(self):
        """Identity + version of every mapping-network tensor: a cached avg_latent is only valid for these weights."""
        return tuple((p.data_ptr(), p._version) for net in (self.neural_field_mapping_network, self.synthesis_mapping_network)
                     for p in net.parameters())

    def generate_avg_latent(self):
        """Mean freq / phase / styles over 10 000 random latents (reference :182-194)."""
        z = torch.randn((10000, self.latent_dim), device=self.neural_field.device)
        fr, ph = self.neural_field_mapping_network(z)
        _, st = self.synthesis_mapping_network(z)
        self.avg_latent = (z.mean(dim=0, keepdim=True), fr.mean(dim=0, keepdim=True), ph.mean(dim=0, keepdim=True),
                           st.mean(dim=0, keepdim=True))
        self._avg_latent_key = self._mapping_key()
        return self.avg_latent

    def cached_avg_latent(self):
        """The cached avg_latent if it was computed from the CURRENT mapping-network weights (load_state_dict, .to() or an
        optimizer step invalidate it), else None."""
        if self.avg_latent is not None and getattr(self, "_avg_latent_key", None) == self._mapping_key():
            return self.avg_latent
        return None

    @torch.no_grad()
    def get_geo_features(self, points, skeletons, vertices, tpose_vertices, fk_matrices, lbs_weights, **kw):
        if self.disable_modulation:
            return torch.zeros(list(points.shape)[:2] + [self.geo_feature_dim], dtype=points.dtype, device=points.device)
        return smpl.get_geo_features(points, skeletons, vertices, tpose_vertices, fk_matrices, lbs_weights,
                                     self.legacy_mode, **kw)

    # ------------------------------------------------------------------ synthesis plan (static packing)
    def synthesis_plan(self, device):
        """Packed synthesis weights for `device` ("cuda", "cuda:0" and torch.device("cuda", 0) name the same plan: an
        engine override set through one spelling must be seen by forward(), which asks with the tensors' device)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        sd = self.state_dict()
        key = (str(device),) + tuple((v.data_ptr(), v._version) for n, v in sd.items()
                                     if n.startswith(("synthesis_network", "synthesis_input")))
        if self._plan is None or self._plan_key != key:
            sn = self.synthesis_network
            self._plan = SynthesisPlan(sd, "synthesis_network", "synthesis_input", sn.num_blocks, sn.mod_blocks,
                                       sn.map3d_mode, device)
            self._plan_key = key
        return self._plan

    # ------------------------------------------------------------------ stages
    def _mapping(self, latent, kwargs):
        with stage(self, "mapping"):
            return self._mapping_impl(latent, kwargs)

    def _mapping_impl(self, latent, kwargs):
        if kwargs.get("neural_field_latent_input", True):
            fr, ph = self.neural_field_mapping_network(latent)
        else:
            fr, ph = self.neural_field_mapping_network(torch.zeros_like(latent))
        _, styles = self.synthesis_mapping_network(latent)
        return fr, ph, styles

    def render(self, *args, differentiable=False, **kwargs):
        """reference :381-523.  -> rgb_render [B,3,Hr,Wr], feature_maps [B,R,F] (channels LAST: it feeds the
        synthesis kernel directly; use .transpose for NCHW), depths [B,R,1], weights [B,R,S,1], None.

        ``staged``/``max_points`` chunking is a memory work-around of the reference and is not needed: the field
        tensor never materialises in the fused path.  ``differentiable=True`` evaluates the field as library GEMMs + HIP
        activation / integration kernels with hand-written adjoints (lib/generators/differentiable.py) instead of the fused
        inference kernel."""
        if differentiable:
            return self._render(*args, differentiable=True, **kwargs)
        with torch.no_grad():
            return self._render(*args, **kwargs)

    def _render(self, freq, phase, conditions, render_width, render_height, ray_start, ray_end, coarse_steps, fine_steps=None,
                h_stddev=0, v_stddev=0, h_mean=0, v_mean=0, hierarchical_sample=False, sample_dist=None,
                lock_view_dependence=False, staged=False, max_points=50000, jitter=None, noise=None, fused=True,
                differentiable=False, **kwargs):
        if hierarchical_sample and differentiable:
            raise NotImplementedError("hierarchical_sample=True has no differentiable path (no shipped config trains with it)")
        nf = self.neural_field
        if (differentiable and not torch.is_grad_enabled() and not hierarchical_sample and fused and self.train_field in ("x3", "x2")
                and nf.device_pack and nf.precision in ("f16x2", "f16x3") and nf.fused_supported(int(coarse_steps))):
            keep, nf.precision = nf.precision, "f16x3" if self.train_field == "x3" else "f16x2"
            try:
                return self._render(freq, phase, conditions, render_width, render_height, ray_start, ray_end, coarse_steps, fine_steps,
                                    h_stddev, v_stddev, h_mean, v_mean, False, sample_dist, lock_view_dependence, staged, max_points,
                                    jitter, noise, fused, differentiable=False, **kwargs)
            finally:
                nf.precision = keep
        if hierarchical_sample:
            return self._render_hierarchical(freq, phase, conditions, render_width, render_height, ray_start, ray_end,
                                             int(coarse_steps), int(coarse_steps if fine_steps is None else fine_steps),
                                             lock_view_dependence, jitter, noise, sample_dist=sample_dist, **kwargs)
        c = conditions
        dev = freq.device
        B, S = freq.shape[0], int(coarse_steps)
        R = render_width * render_height
        focals = c["intrinsics"][:, 0, 0]
        scales = c["scales"].float()
        # RNG order of the reference: jitter U(0,1) [B,R,S,1]; the discarded camera sample; integration noise randn
        with stage(self, "ray_setup"):
            pts, z_vals = vr.sample_rays(focals, scales, c["cam2world_matrices"], S, (render_width, render_height),
                                         ray_start, ray_end, jitter=jitter, perturb=True)
        _consume_camera_rng(sample_dist, B, dev)
        nerf_noise = kwargs.get("nerf_noise", 0)
        if noise is None:
            drawn = torch.randn((B, R, S, 1), device=dev)                          # volume_rendering.py:24
            noise = drawn * nerf_noise if nerf_noise != 0 else None
        can_fuse = fused and not differentiable and self.neural_field.fused_supported(S)
        # A4 inside the fused render (round 4): only the nearest-vertex search runs as its own kernel, the features are built in
        # the field kernel's prologue -- the [B, N, 31] tensor is never written (H3D_FUSE_GEO=0: the two-kernel path)
        geo_in = can_fuse and self.fuse_geo and self.neural_field.render_geo_supported(S)
        with stage(self, "geo_features"):
            if geo_in:
                vik = smpl.vertex_inverse_transforms(c["fk_matrices"], c["lbs_weights"])
                nn_index = smpl.nearest_vertex(pts, c["vertices"], ray_shape=(render_height, render_width, S))
                geo = None
            else:
                geo = self.get_geo_features(pts, c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"],
                                            c["lbs_weights"])
        dirs = None
        if not lock_view_dependence:
            dirs = vr.ray_directions_world(focals, c["cam2world_matrices"], (render_width, render_height), S)
        scaler = 2.0 / self.side_length
        clamp_mode = kwargs["clamp_mode"]
        last_back, white_back = kwargs.get("last_back", False), kwargs.get("white_back", False)
        if geo_in:
            with stage(self, "render_fused"):
                feats, depths, weights = self.neural_field.render_geo(
                    pts, freq, phase, nn_index, c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], vik, dirs, z_vals, S,
                    legacy_mode=self.legacy_mode, input_scaler=scaler, noise=noise, clamp_mode=clamp_mode,
                    last_back=last_back, white_back=white_back)
        elif differentiable:
            with stage(self, "neural_field"):
                field = field_forward(self.neural_field, pts, freq, phase, geo, dirs, input_scaler=scaler)
            with stage(self, "ray_integrate"):
                feats, depths, weights = vr.ray_integration(field.reshape(B, R, S, -1), z_vals, noise_std=0, noise=noise,
                                                            clamp_mode=clamp_mode, last_back=last_back,
                                                            white_back=white_back, consume_rng=False)
        elif can_fuse:
            with stage(self, "render_fused"):
                feats, depths, weights = self.neural_field.render(pts, freq, phase, geo, dirs, z_vals, S,
                                                                  input_scaler=scaler, noise=noise, clamp_mode=clamp_mode,
                                                                  last_back=last_back, white_back=white_back)
        else:
            with stage(self, "neural_field"):
                field = self.neural_field(pts, freq, phase, geo, dirs, input_scaler=scaler, differentiable=False)
            with stage(self, "ray_integrate"):
                feats, depths, weights = vr.ray_integration(field.reshape(B, R, S, -1), z_vals, noise_std=0, noise=noise,
                                                            clamp_mode=clamp_mode, last_back=last_back,
                                                            white_back=white_back, consume_rng=False)
        rgb_render = (feats[..., :3] * 2 - 1).reshape(B, render_height, render_width, 3).permute(0, 3, 1, 2)
        return rgb_render, feats[..., 3:], depths, weights, None

    def _render_hierarchical(self, freq, phase, c, render_width, render_height, ray_start, ray_end, S, Sf,
                             lock_view_dependence, jitter, noise, noise_coarse=None, fine_u=None, sample_dist=None,
                             **kwargs):
        """reference :449-516: coarse pass -> compositing weights -> importance samples -> fine pass -> merge by depth ->
        integration over all samples.  The field tensors materialise here (unfused kernels); random tensors can be
        injected (jitter, noise_coarse [B,R,S,1], fine_u [B*R,Sf], noise [B,R,S+Sf,1]) or are drawn where the reference
        draws them."""
        dev = freq.device
        B, R = freq.shape[0], render_width * render_height
        focals, scales = c["intrinsics"][:, 0, 0], c["scales"].float()
        res = (render_width, render_height)
        with stage(self, "ray_setup"):
            pts, z_vals = vr.sample_rays(focals, scales, c["cam2world_matrices"], S, res, ray_start, ray_end, jitter=jitter,
                                         perturb=True)
            origins, ray_dirs = vr.ray_frame_world(focals, c["cam2world_matrices"], res)
        _consume_camera_rng(sample_dist, B, dev)
        nerf_noise = kwargs.get("nerf_noise", 0)
        clamp_mode = kwargs["clamp_mode"]
        scaler = 2.0 / self.side_length
        mesh = (c["skeletons_xyz"], c["vertices"], c["tpose_vertices"], c["fk_matrices"], c["lbs_weights"])

        def dirs_for(n_steps):
            if lock_view_dependence:
                return None
            return ray_dirs.unsqueeze(2).expand(B, R, n_steps, 3).reshape(B, R * n_steps, 3).contiguous()

        with stage(self, "geo_features"):
            geo = self.get_geo_features(pts, *mesh)
        with stage(self, "neural_field"):
            coarse = self.neural_field(pts, freq, phase, geo, dirs_for(S), input_scaler=scaler,
                                       differentiable=False).reshape(B, R, S, -1)
        if noise_coarse is None:
            drawn = torch.randn((B, R, S, 1), device=dev)                          # volume_rendering.py:24, first call
            noise_coarse = drawn * nerf_noise if nerf_noise != 0 else None
        with stage(self, "ray_integrate"):
            _, _, w = vr.ray_integration(coarse, z_vals, noise_std=0, noise=noise_coarse, clamp_mode=clamp_mode,
                                         consume_rng=False)
        with stage(self, "resample"):
            w = w.reshape(B * R, S) + 1e-5
            zv = z_vals.reshape(B * R, S)
            z_mid = 0.5 * (zv[:, :-1] + zv[:, 1:])
            fine_z = vr.sample_pdf(z_mid, w[:, 1:-1], Sf, det=False, u=fine_u).reshape(B, R, Sf, 1)
            fine_pts = vr.ray_points(origins, ray_dirs, fine_z)
        with stage(self, "geo_features"):
            geo = self.get_geo_features(fine_pts, *mesh)
        with stage(self, "neural_field"):
            fine = self.neural_field(fine_pts, freq, phase, geo, dirs_for(Sf), input_scaler=scaler,
                                     differentiable=False).reshape(B, R, Sf, -1)
        with stage(self, "resample"):
            all_out, all_z = vr.merge_samples(fine, coarse, fine_z, z_vals)
        if noise is None:
            drawn = torch.randn((B, R, S + Sf, 1), device=dev)                     # volume_rendering.py:24, second call
            noise = drawn * nerf_noise if nerf_noise != 0 else None
        with stage(self, "ray_integrate"):
            feats, depths, weights = vr.ray_integration(all_out, all_z, noise_std=0, noise=noise, clamp_mode=clamp_mode,
                                                        last_back=kwargs.get("last_back", False),
                                                        white_back=kwargs.get("white_back", False), consume_rng=False)
        rgb_render = (feats[..., :3] * 2 - 1).reshape(B, render_height, render_width, 3).permute(0, 3, 1, 2)
        return rgb_render, feats[..., 3:], depths, weights, None

    def _synthesize(self, feature_maps, styles, render_hw, differentiable=False):
        if differentiable:
            with stage(self, "synthesis"):
                return synthesis_forward(self, feature_maps, styles, render_hw, (self.gen_height, self.gen_width),
                                         training=self.training, group=getattr(self, "process_group", None))
        plan = self.synthesis_plan(feature_maps.device)
        return plan.run(feature_maps, styles.reshape(styles.shape[0], -1), render_hw, (self.gen_height, self.gen_width),
                        owner=self)

    def wants_autograd(self, kwargs):
        """Which evaluation a forward call gets.  ``.train()`` mode -> the differentiable path with the reference's train-mode
        semantics (batch-statistics BatchNorm, spectral-norm power iteration), whether or not autograd is recording;
        ``.eval()`` mode -> the fused inference engines, never recorded.  ``differentiable=True / False`` overrides the
        choice (True in eval mode: gradients through the running-statistics network, e.g. latent optimisation)."""
        d = kwargs.get("differentiable")
        return self.training if d is None else bool(d)

    def forward(self, latent, conditions, render_height, render_width, latent_indices=None, **kwargs):
        """reference :208-280 -> {"rgbs", "rgbs_render"}"""
        if not latent.is_cuda:
            _lib.need_cuda(latent)
        diff = self.wants_autograd(kwargs)
        kwargs.pop("differentiable", None)
        with torch.cuda.device(latent.device):          # kernels launch on the current device's stream
            if diff:
                return self._forward(latent, conditions, render_height, render_width, latent_indices, differentiable=True,
                                     **kwargs)
            with torch.no_grad():
                return self._forward(latent, conditions, render_height, render_width, latent_indices, **kwargs)

    def _forward(self, latent, conditions, render_height, render_width, latent_indices=None, differentiable=False, **kwargs):
        if kwargs.get("disable_render", False):
            raise NotImplementedError("disable_render=True is not set by any config and has no HIP path")
        num_steps = kwargs.get("num_steps", 24)
        if latent_indices is not None:
            latent = self.latent_pool(latent_indices)
        fr, ph, styles = self._mapping(latent, kwargs)
        rk = {k: v for k, v in kwargs.items() if k not in ("coarse_steps", "fine_steps", "render_width", "render_height")}
        rgb_render, fmap, _, _, _ = self.render(fr, ph, conditions, render_width, render_height,
                                                coarse_steps=num_steps, fine_steps=num_steps, differentiable=differentiable,
                                                **rk)
        if kwargs.get("disable_synthesis", False):
            return {"rgbs": rgb_render, "rgbs_render": rgb_render}
        rgb = self._synthesize(fmap, styles, (render_height, render_width), differentiable)
        return {"rgbs": rgb, "rgbs_render": rgb_render}

    @torch.no_grad()
    def staged_forward(self, latent, conditions, render_height, render_width, truncation_psi, **kwargs):
        """reference :282-378 -> {"rgbs", "rgbs_render", "depths" (CPU, as the reference), "skeletons"}.
        ``avg_latent=`` kwarg (or a cached self.avg_latent with cache_avg_latent=True) skips the 10 000-sample pass."""
        if not latent.is_cuda:
            _lib.need_cuda(latent)
        with torch.cuda.device(latent.device):
            return self._staged_forward(latent, conditions, render_height, render_width, truncation_psi, **kwargs)

    def _staged_forward(self, latent, conditions, render_height, render_width, truncation_psi, **kwargs):
        if kwargs.get("disable_render", False):
            raise NotImplementedError("disable_render=True is not set by any config and has no HIP path")
        num_steps = kwargs.get("num_steps", 24)
        B = latent.shape[0]
        fr, ph, styles = self._mapping(latent, kwargs)
        if truncation_psi < 1.0:
            avg = kwargs.get("avg_latent")
            if avg is None:
                avg = self.cached_avg_latent() if kwargs.get("cache_avg_latent", False) else None
                if avg is None:
                    avg = self.generate_avg_latent()
            az, af, ap, ast = avg
            fr = af + truncation_psi * (fr - af)
            ph = ap + truncation_psi * (ph - ap)
            latent = az + truncation_psi * (latent - az)
            styles = ast + truncation_psi * (styles - ast)
        rk = {k: v for k, v in kwargs.items() if k not in ("coarse_steps", "fine_steps", "render_width", "render_height",
                                                           "staged", "avg_latent", "cache_avg_latent")}
        rgb_render, fmap, depths, _, _ = self.render(fr, ph, conditions, render_width, render_height,
                                                     coarse_steps=num_steps, fine_steps=num_steps, staged=True, **rk)
        if kwargs.get("disable_synthesis", False):
            from ..components.resample import bilinear_resize
            out = {"rgbs": bilinear_resize(rgb_render.contiguous(), (self.gen_height, self.gen_width)),
                   "rgbs_render": rgb_render}
        else:
            out = {"rgbs": self._synthesize(fmap, styles, (render_height, render_width)), "rgbs_render": rgb_render}
        zc = conditions["intrinsics"][:, 0, 0] / conditions["scales"].float()
        depth = ((depths - zc.view(B, 1, 1)) / (kwargs["depth_length"] / 2.0)).clamp(-1.0, 1.0)
        depth_map = depth.reshape(B, render_height, render_width).unsqueeze(1).contiguous()
        out.update({"depths": depth_map if kwargs.get("keep_depth_on_device", False) else depth_map.cpu(),
                    "skeletons": conditions["skeletons_xyz"]})
        return out
