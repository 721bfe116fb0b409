"""Fused bias + activation with the reference's Python signature (lib/components/ops/bias_act.py:52-86), executed by the HIP
kernels behind h3d_bias_act / h3d_bias_act_grad.  Differentiable to second order, like the reference's custom-op path
(bias_act.py:126-207): first-order backward = one fused kernel, and that kernel is itself differentiable (R1-style double
backward)."""
import math

import torch

from .... import _lib

_S2 = math.sqrt(2.0)
# name -> (default alpha, default gain, kernel index, what the derivative is evaluated from, has a second derivative);
# table of reference bias_act.py:20-31
activation_funcs = {
    "linear": (0.0, 1.0, 1, "", False), "relu": (0.0, _S2, 2, "y", False), "lrelu": (0.2, _S2, 3, "y", False),
    "tanh": (0.0, 1.0, 4, "y", True), "sigmoid": (0.0, 1.0, 5, "y", True), "elu": (0.0, 1.0, 6, "y", True),
    "selu": (0.0, 1.0, 7, "y", True), "softplus": (0.0, 1.0, 8, "y", True), "swish": (0.0, _S2, 9, "x", True),
}
_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def _bias_layout(x, b, dim):
    if b is None:
        return None, 1, 1
    assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    return b.to(x.dtype).contiguous(), b.shape[0], (x.stride(dim) if x.shape[dim] > 1 else 1)


def _forward(x, b, dim, idx, alpha, gain, clamp):
    bc, size_b, step_b = _bias_layout(x, b, dim)
    y = torch.empty_like(x)
    rc = _lib.load().h3d_bias_act(_lib.ptr(x), _lib.ptr(bc), _lib.ptr(y), x.numel(), _DTYPES[x.dtype], size_b, step_b, idx,
                                  alpha, gain, clamp, _lib.stream_handle())
    _lib.check(rc, "h3d_bias_act")
    return y


def _grad(order, g, b, xref, yref, dy2, dim, idx, alpha, gain, clamp):
    g = g.contiguous()
    dy2 = None if dy2 is None else dy2.contiguous()          # bound to a local: it must outlive the launch
    bc, size_b, step_b = _bias_layout(g, b, dim)
    out = torch.empty_like(g)
    rc = _lib.load().h3d_bias_act_grad(_lib.ptr(g), _lib.ptr(bc), _lib.ptr(xref), _lib.ptr(yref),
                                       _lib.ptr(dy2), _lib.ptr(out), g.numel(),
                                       _DTYPES[g.dtype], size_b, step_b, order, idx, alpha, gain, clamp, _lib.stream_handle())
    _lib.check(rc, "h3d_bias_act_grad")
    return out


def _other_dims(t, dim):
    return [i for i in range(t.ndim) if i != dim]


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, spec, alpha, gain, clamp):
        _, _, idx, ref, second = spec
        y = _forward(x, b, dim, idx, alpha, gain, clamp)
        # x is read by the kernels for swish only; for the other twice-differentiable activations it is kept as the graph
        # anchor the second-order gradient is returned on (d(dx)/dx, evaluated from y), as the reference does
        keep_x = ref == "x" or second
        # y feeds the derivative of the "y" activations and the clamp mask (the reference drops the mask for 'linear', whose
        # gradient then ignores the clamp; here the clamped elements get zero gradient for every activation)
        keep_y = ref == "y" or (clamp >= 0 and ref != "x")
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if keep_y else None)
        ctx.cfg = (dim, spec, alpha, gain, clamp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dim, spec, alpha, gain, clamp = ctx.cfg
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if spec[2] != 1 or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, dim, spec, alpha, gain, clamp)
        if ctx.needs_input_grad[1]:
            db = dx.sum(_other_dims(dx, dim))
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, dim, spec, alpha, gain, clamp):
        dx = _grad(1, dy, b, x, y, None, dim, spec[2], alpha, gain, clamp)
        ctx.save_for_backward(dy if spec[4] else None, x, b, y)
        ctx.cfg = (dim, spec, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        dim, spec, alpha, gain, clamp = ctx.cfg
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, spec, alpha, gain, clamp)
        if spec[4] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _grad(2, d_dx, b, x, y, dy, dim, spec[2], alpha, gain, clamp)
            if ctx.needs_input_grad[2]:
                d_b = d_x.sum(_other_dims(d_x, dim))
        return d_dy, d_x, d_b, None, None, None, None, None, None


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="hip"):
    """y = clamp(act(x + b) * gain).  ``impl`` is accepted for signature compatibility ('ref'/'cuda' in the
    reference); every value routes to the HIP kernels -- there is no PyTorch fallback in this package."""
    assert isinstance(x, torch.Tensor)
    if act not in activation_funcs:
        raise KeyError(f"unknown activation {act!r}")
    if x.dtype == torch.bfloat16:                     # bf16 autocast: the kernels take f32 / f16 / f64
        x = x.float()
    if x.dtype not in _DTYPES:
        raise TypeError(f"bias_act: unsupported dtype {x.dtype}")
    _lib.need_cuda(x, b)
    spec = activation_funcs[act]
    alpha = float(spec[0] if alpha is None else alpha)
    gain = float(spec[1] if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    assert clamp == -1 or clamp >= 0
    xc = x if x.is_contiguous() else x.contiguous()     # dense, canonical strides
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    if torch.is_grad_enabled() and (xc.requires_grad or (b is not None and b.requires_grad)):
        return _BiasAct.apply(xc, b, dim, spec, alpha, gain, clamp)
    return _forward(xc, b, dim, spec[2], alpha, gain, clamp)
