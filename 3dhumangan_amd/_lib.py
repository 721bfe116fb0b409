"""ctypes binding of libh3d.so (the C ABI declared in include/h3d.h).

There is deliberately no CPU fallback here: if the HIP library is missing or a call fails, the
caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libh3d.so")
LIB_PATH = os.environ.get("H3D_LIB", LIB_PATH)       # development: an experimental build of the same library

_lib = None


class H3DError(RuntimeError):
    pass


class FieldParams(C.Structure):
    _fields_ = [("w_coord", C.c_void_p), ("b_coord", C.c_void_p), ("w_geo", C.c_void_p), ("b_geo", C.c_void_p),
                ("w_film", C.c_void_p * 4), ("b_film", C.c_void_p * 4), ("w_sigma", C.c_void_p),
                ("b_sigma", C.c_void_p), ("w_color", C.c_void_p), ("b_color", C.c_void_p), ("w_rgb", C.c_void_p),
                ("b_rgb", C.c_void_p), ("w_feat", C.c_void_p), ("b_feat", C.c_void_p)]


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    "h3d_version": (C.c_int, []),
    "h3d_last_error": (C.c_char_p, []),
    "h3d_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, _i]),
    "h3d_ray_integrate": (C.c_int, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_ray_setup": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "h3d_geo_features": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _p]),
    "h3d_nearest_vertex": (C.c_int, [_p, _p, _p, _i, _l, _i, _p]),
    "h3d_mesh_sort_bytes": (C.c_int64, [_i, _i]),
    "h3d_mesh_sort": (C.c_int, [_p, _p, _i, _i, _p]),
    "h3d_geo_features_sorted": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _p]),
    "h3d_nearest_vertex_sorted": (C.c_int, [_p, _p, _p, _i, _l, _i, _p]),
    "h3d_nearest_vertex_sorted_rays": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "h3d_render_fused_x2_geo": (C.c_int, [_p] * 7 + [_i, _i] + [_p] * 8 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _p]),
    "h3d_render_fused_x3_geo": (C.c_int, [_p] * 7 + [_i, _i] + [_p] * 8 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _p]),
    "h3d_render_fused_x2_geo_ref": (C.c_int, [_p] * 7 + [_i, _i] + [_p] * 8 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _f, _f, _p, _p, _i, _p]),
    "h3d_render_fused_x3_geo_units": (C.c_int, [_p] * 7 + [_i, _i] + [_p] * 8 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _p, _p, _i, _p]),
    "h3d_field_pack_size": (C.c_int64, [_i, _i]),
    "h3d_field_pack": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p]),
    "h3d_neural_field": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _f, _p]),
    "h3d_render_fused": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i,
                                   _i, _p]),
    "h3d_field_pack_x3_size": (C.c_int64, [_i, _i]),
    "h3d_field_pack_x3": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p]),
    "h3d_field_pack_x3_device": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p, _p]),
    "h3d_field_x3_layout": (C.c_int, [_i, _i, C.POINTER(C.c_int64), _i]),
    "h3d_neural_field_x3": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _f, _p]),
    "h3d_render_fused_x3": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i,
                                      _i, _i, _p]),
    "h3d_conv_x3_tiling": (C.c_int, [_i, _i, C.POINTER(C.c_int)]),
    "h3d_conv_x3_pack": (C.c_int, [_p, _p, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_pack_f16": (C.c_int, [_p, _p, _i, _i, _i, _i, _p]),
    "h3d_conv_x3": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_f16": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_pack_f16x1": (C.c_int, [_p, _p, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_f16x1": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_wgrad_x3_f16": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_wgrad_x3_bias": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_wgrad_x3_bias_f16": (C.c_int, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_wgrad_x3_slices": (C.c_int, [_i, _i, _i, _i, _i, _i]),
    "h3d_conv_wgrad_x3_fused": (C.c_int, [_i, _i, _i, _i, _i]),
    "h3d_conv_wgrad_x3": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_field_pack_x2_size": (C.c_int64, [_i, _i]),
    "h3d_field_pack_x2": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p]),
    "h3d_field_pack_x2_device": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p, _p]),
    "h3d_field_x2_layout": (C.c_int, [_i, _i, C.POINTER(C.c_int64), _i]),
    "h3d_neural_field_x2": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _f, _p]),
    "h3d_render_fused_x2": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i,
                                      _i, _i, _p]),
    "h3d_field_pack_x3t_size": (C.c_int64, [_i, _i]),
    "h3d_field_pack_x3t": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p]),
    "h3d_field_pack_x2t": (C.c_int, [C.POINTER(FieldParams), _i, _i, _p]),
    "h3d_field_x3t_layout": (C.c_int, [_i, _i, C.POINTER(C.c_int64), _i]),
    "h3d_neural_field_x3t": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _f, _p]),
    "h3d_render_fused_x3t": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i,
                                       _i, _i, _p]),
    "h3d_neural_field_x3t_tier": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _f, _i, _p]),
    "h3d_render_fused_x3t_tier": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i,
                                            _i, _i, _i, _p]),
    "h3d_pack_matrix": (C.c_int, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_synthesis": (C.c_int, [_p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "h3d_synthesis_x3": (C.c_int, [_p, _l, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p]),
    "h3d_synthesis_x2": (C.c_int, [_p, _l, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p]),
    "h3d_synthesis_x2_extra_lds": (C.c_int, [_i]),
    "h3d_synthesis_x2_guarded": (C.c_int, [_p, _l, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _p]),
    "h3d_synthesis_x3_if": (C.c_int, [_p, _l, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _p]),
    "h3d_synthesis_x3_tiles": (C.c_int, [_p, _l, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "h3d_synthesis_check": (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "h3d_synthesis_x3t_tiles": (C.c_int, [_i]),
    "h3d_synthesis_x3t": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "h3d_synthesis_x3t_tier": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "h3d_synthesis_x3t_tier_guarded": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "h3d_modconv1x1": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _f, _p]),
    "h3d_modconv2d": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_synthesis_x3_geometry_ok": (C.c_int, [_i, _i, _i, _i]),
    "h3d_synthesis_x3_lds_bytes": (C.c_int64, [_i, _i, _i, _i, _i]),
    "h3d_sample_pdf": (C.c_int, [_p, _p, _p, _p, _l, _i, _i, _f, _p]),
    "h3d_ray_points": (C.c_int, [_p, _p, _p, _p, _i, _l, _i, _p]),
    "h3d_merge_samples": (C.c_int, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "h3d_bilinear_resize": (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_bilinear_resize_cl": (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_bilinear_resize_cl_bwd": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_bilinear_resize_cl_relu": (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_bilinear_resize_cl_relu_bwd": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_film_sin_rows": (C.c_int, []),
    "h3d_film_sin": (C.c_int, [_p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_film_sin_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_ray_integrate_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_spectral_norm_scratch": (C.c_int64, [_i, _i]),
    "h3d_spectral_norm": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "h3d_spectral_norm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "h3d_up2_mask": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p]),
    "h3d_pool2_mask": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p]),
    "h3d_wgrad_narrow_rows": (C.c_int, []),
    "h3d_wgrad_narrow": (C.c_int, [_p, _p, _p, _l, _i, _i, _i, _p]),
    "h3d_wgrad_narrow_f16": (C.c_int, [_p, _p, _p, _l, _i, _i, _i, _p]),
    "h3d_wgrad_narrow_sums": (C.c_int, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_wgrad_x3_slices": (C.c_int, [_l, _i, _i]),
    "h3d_wgrad_x3": (C.c_int, [_p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_wgrad_x3_bias": (C.c_int, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "h3d_wgrad_reduce": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_add": (C.c_int, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_ex": (C.c_int, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_conv_x3_slices": (C.c_int, [_i, _i, _i, _l, _i]),
    "h3d_conv_x3_moment_rows": (C.c_int, []),
    "h3d_conv_x3_nt_for": (C.c_int, [_i, _i, _l]),
    "h3d_conv_x3_pack_nt": (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "h3d_pad_channels_cl": (C.c_int, [_p, _p, _i, _i, _i, _l, _l, _l, _l, _i, _p]),
    "h3d_spade_rows": (C.c_int, []),
    "h3d_channel_moments": (C.c_int, [_p, _p, _i, _l, _i, _p]),
    "h3d_spade_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_spade_bwd_reduce": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_spade_bwd_apply": (C.c_int, [_p] * 14 + [_i, _l, _i, _i, _f, _p]),
    "h3d_channel_moments_f16": (C.c_int, [_p, _p, _i, _l, _i, _p]),
    "h3d_spade_fwd_f16": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_spade_bwd_reduce_f16": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _f, _p]),
    "h3d_spade_bwd_apply_f16": (C.c_int, [_p] * 14 + [_i, _l, _i, _i, _f, _p]),
    "h3d_spade_bwd_apply_acc": (C.c_int, [_i] + [_p] * 16 + [_i, _l, _i, _i, _f, _p]),
    "h3d_rows_sum_f64": (C.c_int, [_p, _p, _l, _i, _p]),
    "h3d_bn_finish": (C.c_int, [_p] * 7 + [_i, _f, _f, _p]),
    "h3d_bn_bwd_finish": (C.c_int, [_p] * 7 + [_i, _p]),
    "h3d_bias_act": (C.c_int, [_p, _p, _p, _l, _i, _l, _l, _i, _f, _f, _f, _p]),
    "h3d_bias_act_grad": (C.c_int, [_p, _p, _p, _p, _p, _p, _l, _i, _l, _l, _i, _i, _f, _f, _f, _p]),
    "h3d_upfirdn2d": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_l), _i, _i, _i, _i, C.POINTER(_l),
                                _i, _i, _i, _i, _i, _i, _i, _f, _p]),
}


def load():
    """Load (once) and return the ctypes handle.  Raises H3DError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise H3DError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  -- load torch's HIP runtime first so libh3d binds to the same libamdhip64
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().h3d_last_error().decode(errors="replace")
        raise H3DError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def need_cuda(*tensors):
    """The product has no CPU path: refuse CPU tensors loudly.  All tensors of a call must live on ONE device and that
    device must be the current one: the kernels launch on the current device's current stream (stream_handle), so a
    tensor from another GPU would be read through a foreign pointer with no ordering against the torch ops that
    produced it.  Map3DGenerator.forward enters ``torch.cuda.device(latent.device)`` itself; direct callers of the op
    wrappers do the same."""
    import torch
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise H3DError("3dhumangan_amd ops run on a ROCm device only (got a CPU tensor); there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise H3DError(f"3dhumangan_amd op called with tensors on {dev} and {t.device}: all operands must share a device")
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        raise H3DError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                       f"wrap the call in `with torch.cuda.device({dev.index}):`")


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def aligned16(t):
    """`t` itself when its first element sits on a 16-byte boundary, else a copy (fresh allocations are aligned).  The kernels
    that read 16 bytes per lane require it (H3D_REQUIRE(aligned16)); a contiguous view at an odd storage offset -- a slice of a
    bias vector, a row range of a larger buffer -- is contiguous already, so .contiguous() alone does not fix it."""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone()
