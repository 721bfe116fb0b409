"""COORDCONCATSIREN -- the pluggable pose-conditioned implicit function (reference:
lib/implicit_funcitions/modulated.py:6-75), evaluated by the fp32-MFMA HIP kernel behind
h3d_neural_field / h3d_render_fused.

Same constructor arguments, same parameter names (so reference state_dicts load unchanged), same forward
signature and output channel order [rgb(3), feat(F), sigma(1)].  `.eval()`: the fused inference kernels (no autograd
through them).  `.train()` (or ``differentiable=True``): the differentiable evaluation of lib/generators/differentiable.py --
library GEMMs + the HIP film_sin kernels with hand-written adjoints.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from ... import _lib


class _Dense(nn.Module):
    """Holder that reproduces the reference's `<name>.layer.{weight,bias}` parameter paths."""

    def __init__(self, n_in, n_out):
        super().__init__()
        self.layer = nn.Linear(n_in, n_out)


def _uniform_(linear, bound):
    with torch.no_grad():
        linear.weight.uniform_(-bound, bound)


class COORDCONCATSIREN(nn.Module):

    def __init__(self, input_dim=2, latent_dim=100, hidden_dim=256, geo_feature_dim=88, output_dim=1, feature_dim=32,
                 num_blocks=9, device=None):
        super().__init__()
        if input_dim != 3 or geo_feature_dim != 31 or num_blocks != 4:
            raise NotImplementedError("the HIP field kernel is built for input_dim=3, geo_feature_dim=31, 4 FiLM blocks "
                                      "(every shipped config); got "
                                      f"{input_dim}/{geo_feature_dim}/{num_blocks}")
        self.device = device
        self.input_dim, self.latent_dim, self.hidden_dim = input_dim, latent_dim, hidden_dim
        self.geo_feature_dim, self.output_dim, self.feature_dim = geo_feature_dim, output_dim, feature_dim
        H = hidden_dim
        self.first_layer_coord = _Dense(input_dim, H)
        self.first_layer_mod = _Dense(geo_feature_dim, H)
        self.network = nn.ModuleList([_Dense(2 * H, H)] + [_Dense(H, H) for _ in range(num_blocks - 1)])
        self.sigma_layer = nn.Linear(H, 1)
        self.color_layer_sine = _Dense(H + 3, H)
        self.color_layer_linear = nn.Linear(H, 3)
        self.feature_layer_linear = nn.Linear(H, feature_dim)
        # SIREN initialisation (reference pigan_layers.py:26-53): U(+-sqrt(6/fan_in)/25), first layers U(+-1/fan_in)
        for lin in [d.layer for d in self.network] + [self.sigma_layer, self.color_layer_sine.layer,
                                                      self.color_layer_linear, self.feature_layer_linear]:
            _uniform_(lin, math.sqrt(6.0 / lin.weight.shape[1]) / 25.0)
        for lin in (self.first_layer_coord.layer, self.first_layer_mod.layer):
            _uniform_(lin, 1.0 / lin.weight.shape[1])
        self._packed = {}
        # Arithmetic engine (all meet the 1e-3 parity budget):
        #   "f16x2"   as f16x3 with the two cross products of a contraction in one block-scaled fp6 instruction (widths <= 256)
        #   "f16x3"   split-operand f16 matrix cores, activations register-resident (widths <= 256)
        #   "f16x2t"  the x2 arithmetic on the LDS-resident engine, any width <= 448
        #   "f16x3t"  split-operand f16 matrix cores, activations LDS-resident, any width <= 448 (MAP3DBN 384, MAP3DBN512L 420)
        #   "f32"     fp32 matrix cores (any width <= 512)
        # and, NOT within the 1e-3 budget (the "fp16 MFMA path" tier of BASELINE config 5, ~1e-2 on the render; opt-in):
        #   "f16x1t"  plain f16 matrix-core products on the f16x3t engine (one product instead of three)
        widest = max(hidden_dim, feature_dim)
        default = "f16x2" if widest <= 256 else "f16x2t" if widest <= 448 else "f32"
        self.precision = os.environ.get("H3D_FIELD_PRECISION", default)
        # x2 render: refinement of ill-conditioned last samples on the three-product engine (render_geo; round 6)
        self.refine_last_sample = os.environ.get("H3D_FIELD_REFINE", "1") != "0"
        # weights packed on the device they live on (round 6; H3D_FIELD_PACK=host: the D2H copy + host packer + H2D copy of rounds 1-5)
        self.device_pack = os.environ.get("H3D_FIELD_PACK", "device") != "host"
        self.refine_eps = float(os.environ.get("H3D_FIELD_REFINE_EPS", "1e-3"))
        self.refine_capacity = int(os.environ.get("H3D_FIELD_REFINE_CAP", "128"))       # listed units per batch item
        self._refine_buf = None
        self._sigma_scale_cache = None

    # ---- weight packing (host, once per weight version)
    def _params_for_pack(self):
        return [self.first_layer_coord.layer, self.first_layer_mod.layer] + [d.layer for d in self.network] + \
               [self.sigma_layer, self.color_layer_sine.layer, self.color_layer_linear, self.feature_layer_linear]

    # engine -> (pack-size, pack, field, fused-render) entry points of the C ABI, the sample tile of the fused kernel and
    # the extra trailing arguments (before the stream) of the field / render entry points
    _ENGINES = {
        "f16x3": ("h3d_field_pack_x3_size", "h3d_field_pack_x3", "h3d_neural_field_x3", "h3d_render_fused_x3", 32, ()),
        "f16x2": ("h3d_field_pack_x2_size", "h3d_field_pack_x2", "h3d_neural_field_x2", "h3d_render_fused_x2", 32, ()),
        "f16x3t": ("h3d_field_pack_x3t_size", "h3d_field_pack_x3t", "h3d_neural_field_x3t", "h3d_render_fused_x3t", 64, ()),
        "f16x1t": ("h3d_field_pack_x3t_size", "h3d_field_pack_x3t", "h3d_neural_field_x3t_tier", "h3d_render_fused_x3t_tier",
                   64, (1,)),
        "f16x2t": ("h3d_field_pack_x3t_size", "h3d_field_pack_x2t", "h3d_neural_field_x3t_tier", "h3d_render_fused_x3t_tier",
                   64, (4,)),
        "f32": ("h3d_field_pack_size", "h3d_field_pack", "h3d_neural_field", "h3d_render_fused", 64, ()),
    }

    # packers with a device-side twin (csrc/field_x3.hip: field_pack_kernel)
    _DEVICE_PACKERS = {"h3d_field_pack_x3": "h3d_field_pack_x3_device", "h3d_field_pack_x2": "h3d_field_pack_x2_device"}

    def _engine(self):
        if self.precision not in self._ENGINES:
            raise ValueError(f"unknown precision {self.precision!r}")
        return self._ENGINES[self.precision]

    def fused_supported(self, num_steps):
        """Sample counts the fused field+integration kernel of the active engine accepts."""
        S, tile = int(num_steps), self._engine()[4]
        return (8 <= S <= tile and S & (S - 1) == 0) or (S > tile and S % tile == 0)

    def packed_weights(self, device):
        """Device blob in MFMA fragment order (csrc/field_common.hpp, csrc/field_x3.hip); cached until a
        parameter changes."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        lins = self._params_for_pack()
        size_name, pack_name = self._engine()[:2]
        x3 = pack_name                       # engines that share a packer share the blob
        key = (str(device), x3) + tuple((p.data_ptr(), p._version) for l in lins for p in (l.weight, l.bias))
        hit = self._packed.get(x3)
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        H, F = self.hidden_dim, self.feature_dim
        on_device = (self.device_pack and pack_name in self._DEVICE_PACKERS and device.type == "cuda"
                     and all(p.device == device for l in lins for p in (l.weight, l.bias)))
        if on_device:
            # round 6: packed where the parameters live (one memset + one launch, no synchronisation, bit-identical blob) -- what lets
            # a weight that changes every optimiser step use the fused render (the D step's no-grad generator forward)
            host = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous()) for l in lins]
        else:
            host = [(l.weight.detach().float().cpu().contiguous(), l.bias.detach().float().cpu().contiguous()) for l in lins]
        P = _lib.FieldParams()
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        P.w_coord, P.b_coord = vp(host[0][0]), vp(host[0][1])
        P.w_geo, P.b_geo = vp(host[1][0]), vp(host[1][1])
        for k in range(4):
            P.w_film[k], P.b_film[k] = host[2 + k][0].data_ptr(), host[2 + k][1].data_ptr()
        P.w_sigma, P.b_sigma = vp(host[6][0]), vp(host[6][1])
        P.w_color, P.b_color = vp(host[7][0]), vp(host[7][1])
        P.w_rgb, P.b_rgb = vp(host[8][0]), vp(host[8][1])
        P.w_feat, P.b_feat = vp(host[9][0]), vp(host[9][1])
        size_fn, pack_fn = getattr(lib, size_name), getattr(lib, pack_name)
        nbytes = size_fn(H, F)
        if nbytes <= 0:
            raise _lib.H3DError(f"field engine {self.precision} does not support widths {H}/{F}")
        if on_device:
            with torch.cuda.device(device):
                dev_blob = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
                rc = getattr(lib, self._DEVICE_PACKERS[pack_name])(ctypes.byref(P), H, F, _lib.ptr(dev_blob), _lib.stream_handle())
            _lib.check(rc, self._DEVICE_PACKERS[pack_name])
        else:
            blob = torch.empty((nbytes + 3) // 4, dtype=torch.float32)
            _lib.check(pack_fn(ctypes.byref(P), H, F, ctypes.c_void_p(blob.data_ptr())), "h3d_field_pack")
            dev_blob = blob.to(device)
        self._packed[x3] = (key, dev_blob)
        return dev_blob

    def forward(self, input, frequencies, phase_shifts, geo_feature, ray_directions, input_scaler=1.,
                geo_feature_scaler=1., differentiable=None, **kwargs):
        """input [B,N,3], frequencies/phase_shifts [B,4H], geo_feature [B,N,31], ray_directions [B,N,3] or None
        (None == the lock_view_dependence direction (0,0,-1))  ->  [B,N,F+4]."""
        if self.training if differentiable is None else differentiable:
            from ..generators.differentiable import field_forward
            _lib.need_cuda(input, frequencies, phase_shifts, geo_feature, ray_directions)
            if input.dim() < 3:
                out = field_forward(self, input.unsqueeze(1), frequencies, phase_shifts, geo_feature.unsqueeze(1),
                                    None if ray_directions is None else ray_directions.unsqueeze(1), input_scaler,
                                    geo_feature_scaler)
                return out.squeeze(1)
            return field_forward(self, input, frequencies, phase_shifts, geo_feature, ray_directions, input_scaler,
                                 geo_feature_scaler)
        with torch.no_grad():
            return self._forward_fused(input, frequencies, phase_shifts, geo_feature, ray_directions, input_scaler,
                                       geo_feature_scaler)

    def _forward_fused(self, input, frequencies, phase_shifts, geo_feature, ray_directions, input_scaler=1.,
                       geo_feature_scaler=1.):
        unsq = input.dim() < 3
        if unsq:
            input, geo_feature = input.unsqueeze(1), geo_feature.unsqueeze(1)
            ray_directions = None if ray_directions is None else ray_directions.unsqueeze(1)
        _lib.need_cuda(input, frequencies, phase_shifts, geo_feature, ray_directions)
        B, N, _ = input.shape
        H, F = self.hidden_dim, self.feature_dim
        pts = input.contiguous().float()
        geo = geo_feature if geo_feature_scaler == 1. else geo_feature * geo_feature_scaler
        geo = geo.contiguous().float()
        dirs = None if ray_directions is None else ray_directions.contiguous().float()
        fr, ph = frequencies.contiguous().float(), phase_shifts.contiguous().float()
        assert fr.shape == (B, 4 * H) and ph.shape == (B, 4 * H)
        out = torch.empty((B, N, F + 4), device=pts.device, dtype=torch.float32)
        blob = self.packed_weights(pts.device)
        fn = getattr(_lib.load(), self._engine()[2])
        rc = fn(_lib.ptr(blob), _lib.ptr(pts), _lib.ptr(geo), _lib.ptr(dirs), _lib.ptr(fr),
                                          _lib.ptr(ph), _lib.ptr(out), B, N, H, F, geo.shape[-1], float(input_scaler),
                                          *self._engine()[5], _lib.stream_handle())
        _lib.check(rc, "h3d_neural_field")
        return out.squeeze(1) if unsq else out

    @torch.no_grad()
    def render(self, input, frequencies, phase_shifts, geo_feature, ray_directions, z_vals, num_steps, input_scaler=1.,
               noise=None, clamp_mode="relu", last_back=False, white_back=False):
        """Fused field evaluation + volume integration (reference: COORDCONCATSIREN.forward followed by
        volume_rendering.ray_integration).  input [B,R*S,3] with the S samples of a ray contiguous.
        -> (features [B,R,F+3], depth [B,R,1], weights [B,R,S,1])."""
        _lib.need_cuda(input, frequencies, phase_shifts, geo_feature, ray_directions, z_vals, noise)
        B, N, _ = input.shape
        S = int(num_steps)
        R = N // S
        H, F = self.hidden_dim, self.feature_dim
        pts = input.contiguous().float()
        geo = geo_feature.contiguous().float()
        dirs = None if ray_directions is None else ray_directions.contiguous().float()
        fr, ph = frequencies.contiguous().float(), phase_shifts.contiguous().float()
        z = z_vals.reshape(B, R, S).contiguous().float()
        nz = None if noise is None else noise.reshape(B, R, S).contiguous().float()
        feats = torch.empty((B, R, F + 3), device=pts.device, dtype=torch.float32)
        depth = torch.empty((B, R, 1), device=pts.device, dtype=torch.float32)
        weights = torch.empty((B, R, S, 1), device=pts.device, dtype=torch.float32)
        blob = self.packed_weights(pts.device)
        mode = {"relu": 0, "softplus": 1}[clamp_mode]
        fn = getattr(_lib.load(), self._engine()[3])
        rc = fn(_lib.ptr(blob), _lib.ptr(pts), _lib.ptr(geo), _lib.ptr(dirs), _lib.ptr(fr),
                                          _lib.ptr(ph), _lib.ptr(z), _lib.ptr(nz), _lib.ptr(feats), _lib.ptr(depth),
                                          _lib.ptr(weights), B, R, S, H, F, geo.shape[-1], float(input_scaler), mode,
                                          int(bool(last_back)), int(bool(white_back)), *self._engine()[5], _lib.stream_handle())
        _lib.check(rc, "h3d_render_fused")
        return feats, depth, weights

    # engines whose fused kernel can build the geometry features itself (A4 inside the render, csrc/field_x3.hip GEOIN)
    _GEO_ENGINES = {"f16x2": "h3d_render_fused_x2_geo", "f16x3": "h3d_render_fused_x3_geo"}

    def render_geo_supported(self, num_steps):
        return self.precision in self._GEO_ENGINES and self.fused_supported(num_steps)

    @torch.no_grad()
    def render_geo(self, input, frequencies, phase_shifts, nn_index, skeletons, vertices, tpose_vertices, vertex_ik,
                   ray_directions, z_vals, num_steps, legacy_mode=False, input_scaler=1., noise=None, clamp_mode="relu",
                   last_back=False, white_back=False):
        """`render` with the geometry features built inside the kernel (reference: get_geo_features,
        lib/components/smpl.py:210-249, then COORDCONCATSIREN.forward and ray_integration): instead of geo_feature [B,N,31] it
        takes nn_index [B,N] int32 (smpl.nearest_vertex), skeletons [B,24,3], vertices / tpose_vertices [B,V,3] and
        vertex_ik [B,V,16] (smpl.vertex_inverse_transforms).  -> (features [B,R,F+3], depth [B,R,1], weights [B,R,S,1])."""
        _lib.need_cuda(input, frequencies, phase_shifts, nn_index, skeletons, vertices, tpose_vertices, vertex_ik,
                       ray_directions, z_vals, noise)
        B, N, _ = input.shape
        S = int(num_steps)
        R = N // S
        H, F = self.hidden_dim, self.feature_dim
        pts = input.contiguous().float()
        idx = nn_index.contiguous()
        if idx.dtype != torch.int32 or tuple(idx.shape) != (B, N):
            raise ValueError("nn_index must be int32 [B, N]")
        sk, vt = skeletons.contiguous().float(), vertices.contiguous().float()
        tv, vik = tpose_vertices.contiguous().float(), vertex_ik.contiguous().float()
        if sk.shape[1:] != (24, 3) or vik.shape[1:] != (vt.shape[1], 16) or tv.shape != vt.shape:
            raise ValueError("skeletons [B,24,3], vertices / tpose_vertices [B,V,3], vertex_ik [B,V,16] expected")
        dirs = None if ray_directions is None else ray_directions.contiguous().float()
        fr, ph = frequencies.contiguous().float(), phase_shifts.contiguous().float()
        z = z_vals.reshape(B, R, S).contiguous().float()
        nz = None if noise is None else noise.reshape(B, R, S).contiguous().float()
        feats = torch.empty((B, R, F + 3), device=pts.device, dtype=torch.float32)
        depth = torch.empty((B, R, 1), device=pts.device, dtype=torch.float32)
        weights = torch.empty((B, R, S, 1), device=pts.device, dtype=torch.float32)
        blob = self.packed_weights(pts.device)
        mode = {"relu": 0, "softplus": 1}[clamp_mode]
        lib = _lib.load()
        common = lambda b: (_lib.ptr(b), _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(sk), _lib.ptr(vt), _lib.ptr(tv), _lib.ptr(vik),
                            vt.shape[1], int(bool(legacy_mode)), _lib.ptr(dirs), _lib.ptr(fr), _lib.ptr(ph), _lib.ptr(z), _lib.ptr(nz),
                            _lib.ptr(feats), _lib.ptr(depth), _lib.ptr(weights), B, R, S, H, F, float(input_scaler), mode,
                            int(bool(last_back)), int(bool(white_back)))
        if self.precision == "f16x2" and self.refine_last_sample:
            # Round 6: rays whose LAST sample's density lies within `refine_eps` of zero (relative to the ray's largest density, floored
            # by the density head's weight norm) are listed by the x2 launch and redone by the three-product engine right behind it:
            # the reference's delta = 1e9 on the last sample (lib/generators/volume_rendering.py:21) turns the SIGN of that density
            # into an alpha of 0 or 1, and the x2 arithmetic's ~1e-4 density error must not decide it.  No host synchronisation.
            cap = self.refine_capacity
            buf = self._refine_buf
            if buf is None or buf[0].device != pts.device or buf[0].shape != (B, cap):
                buf = (torch.zeros(B, cap, dtype=torch.int32, device=pts.device), torch.zeros(B, dtype=torch.int32, device=pts.device))
                self._refine_buf = buf
            keep, self.precision = self.precision, "f16x3"
            try:
                blob3 = self.packed_weights(pts.device)           # the same weights in the x3 format (cached per weight version)
            finally:
                self.precision = keep
            scale = self._sigma_scale(pts.device)
            rc = lib.h3d_render_fused_x2_geo_ref(*common(blob), float(self.refine_eps), scale, _lib.ptr(buf[0]), _lib.ptr(buf[1]), cap,
                                                 _lib.stream_handle())
            rc = rc or lib.h3d_render_fused_x3_geo_units(*common(blob3), _lib.ptr(buf[0]), _lib.ptr(buf[1]), cap, _lib.stream_handle())
        else:
            rc = getattr(lib, self._GEO_ENGINES[self.precision])(*common(blob), _lib.stream_handle())
        _lib.check(rc, "h3d_render_fused_geo")
        return feats, depth, weights

    def _sigma_scale(self, device):
        """Euclidean norm of the density head's weights (the scale of a density whose inputs are sines), cached per weight version:
        the floor of the refinement threshold's scale."""
        w = self.sigma_layer.weight
        key = (w.data_ptr(), w._version, str(device))
        if self._sigma_scale_cache is None or self._sigma_scale_cache[0] != key:
            self._sigma_scale_cache = (key, float(w.detach().float().norm()))
        return self._sigma_scale_cache[1]

    def refined_units(self):
        """Per-item number of wave units the LAST render_geo call listed for the three-product refinement (device tensor [B]; reading
        it synchronises), or None."""
        return None if self._refine_buf is None else self._refine_buf[1]
