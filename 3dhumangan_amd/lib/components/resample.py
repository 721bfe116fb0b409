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


def _cl_forward(x, hw, HW):
    B, _, C = x.shape
    out = torch.empty((B, HW[0] * HW[1], C), device=x.device, dtype=torch.float32)
    rc = _lib.load().h3d_bilinear_resize_cl(_lib.ptr(x), _lib.ptr(out), B, hw[0], hw[1], HW[0], HW[1], C, _lib.stream_handle())
    _lib.check(rc, "h3d_bilinear_resize_cl")
    return out


def _cl_adjoint(dy, hw, HW):
    B, _, C = dy.shape
    tmp = torch.empty((B, hw[0] * HW[1], C), device=dy.device, dtype=torch.float32)
    dx = torch.empty((B, hw[0] * hw[1], C), device=dy.device, dtype=torch.float32)
    rc = _lib.load().h3d_bilinear_resize_cl_bwd(_lib.ptr(dy), _lib.ptr(tmp), _lib.ptr(dx), B, hw[0], hw[1], HW[0], HW[1], C,
                                                 _lib.stream_handle())
    _lib.check(rc, "h3d_bilinear_resize_cl_bwd")
    return dx


class _ResizeCL(torch.autograd.Function):
    """The resize and its adjoint are each other's backward (a linear map and its transpose): closed under differentiation."""

    @staticmethod
    def forward(ctx, x, hw, HW, adjoint):
        ctx.geom = (hw, HW, adjoint)
        x = _lib.aligned16(x.contiguous().float())
        return _cl_adjoint(x, hw, HW) if adjoint else _cl_forward(x, hw, HW)

    @staticmethod
    def backward(ctx, dy):
        hw, HW, adjoint = ctx.geom
        return _ResizeCL.apply(dy, hw, HW, not adjoint), None, None, None


def bilinear_resize_cl(x, hw, HW):
    """x [B, h*w, C] channels-last fp32 (C % 4 == 0) -> [B, H*W, C]; align_corners=False; differentiable to any order."""
    _lib.need_cuda(x)
    assert x.dim() == 3 and x.shape[1] == hw[0] * hw[1] and x.shape[2] % 4 == 0
    return _ResizeCL.apply(x, tuple(hw), tuple(HW), False)


class _ResizeReluCL(torch.autograd.Function):
    """relu(resize(x)) in one pass; the backward reads the gradient through the mask of the saved output in the adjoint's first pass
    (h3d_bilinear_resize_cl_relu / _relu_bwd).  First order only (the generator's training path has no double backward)."""

    @staticmethod
    def forward(ctx, x, hw, HW):
        B, _, C = x.shape
        x = _lib.aligned16(x.contiguous().float())
        out = torch.empty((B, HW[0] * HW[1], C), device=x.device, dtype=torch.float32)
        rc = _lib.load().h3d_bilinear_resize_cl_relu(_lib.ptr(x), _lib.ptr(out), B, hw[0], hw[1], HW[0], HW[1], C, _lib.stream_handle())
        _lib.check(rc, "h3d_bilinear_resize_cl_relu")
        ctx.geom = (hw, HW)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        hw, HW = ctx.geom
        B, _, C = out.shape
        dy = _lib.aligned16(dy.contiguous().float())
        tmp = torch.empty((B, hw[0] * HW[1], C), device=dy.device, dtype=torch.float32)
        dx = torch.empty((B, hw[0] * hw[1], C), device=dy.device, dtype=torch.float32)
        rc = _lib.load().h3d_bilinear_resize_cl_relu_bwd(_lib.ptr(dy), _lib.ptr(out), _lib.ptr(tmp), _lib.ptr(dx), B, hw[0], hw[1], HW[0], HW[1],
                                                          C, _lib.stream_handle())
        _lib.check(rc, "h3d_bilinear_resize_cl_relu_bwd")
        return dx, None, None


def bilinear_resize_relu_cl(x, hw, HW):
    """relu(bilinear_resize_cl(x, hw, HW)) as one kernel forward and a masked adjoint backward; fp32 in / out."""
    _lib.need_cuda(x)
    assert x.dim() == 3 and x.shape[1] == hw[0] * hw[1] and x.shape[2] % 4 == 0
    return _ResizeReluCL.apply(x, tuple(hw), tuple(HW))
