"""Bilinear feature-map resize (the F.interpolate call at reference map3d_generator.py:244-245), HIP-backed."""
import torch

from ... import _lib


def bilinear_resize(x, size):
    """x [B,C,h,w] fp32 -> [B,C,H,W], align_corners=False semantics."""
    _lib.need_cuda(x)
    B, C, h, w = x.shape
    H, W = size
    xin = x.contiguous().float()
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    rc = _lib.load().h3d_bilinear_resize(_lib.ptr(xin), _lib.ptr(out), B, C, h, w, H, W, _lib.stream_handle())
    _lib.check(rc, "h3d_bilinear_resize")
    return out
