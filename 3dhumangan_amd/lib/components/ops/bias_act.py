"""Fused bias + activation with the reference's Python signature (lib/components/ops/bias_act.py:52-86),
executed by the HIP kernel behind h3d_bias_act.  Forward only (inference path)."""
import math

import torch

from .... import _lib

_S2 = math.sqrt(2.0)
# name -> (default alpha, default gain, kernel index); table of reference bias_act.py:20-31
activation_funcs = {
    "linear": (0.0, 1.0, 1), "relu": (0.0, _S2, 2), "lrelu": (0.2, _S2, 3), "tanh": (0.0, 1.0, 4),
    "sigmoid": (0.0, 1.0, 5), "elu": (0.0, 1.0, 6), "selu": (0.0, 1.0, 7), "softplus": (0.0, 1.0, 8),
    "swish": (0.0, _S2, 9),
}
_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="hip"):
    """y = clamp(act(x + b) * gain).  ``impl`` is accepted for signature compatibility ('ref'/'cuda' in the
    reference); every value routes to the HIP kernel -- there is no PyTorch fallback in this package."""
    assert isinstance(x, torch.Tensor)
    if act not in activation_funcs:
        raise KeyError(f"unknown activation {act!r}")
    if x.dtype not in _DTYPES:
        raise TypeError(f"bias_act: unsupported dtype {x.dtype}")
    _lib.need_cuda(x, b)
    def_alpha, def_gain, idx = activation_funcs[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    assert clamp == -1 or clamp >= 0
    xc = x if x.is_contiguous() else x.contiguous()     # dense, canonical strides
    size_b, step_b, bc = 1, 1, None
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        bc = b.to(x.dtype).contiguous()
        size_b, step_b = b.shape[0], xc.stride(dim) if xc.shape[dim] > 1 else 1
    y = torch.empty_like(xc)
    rc = _lib.load().h3d_bias_act(_lib.ptr(xc), _lib.ptr(bc), _lib.ptr(y), xc.numel(), _DTYPES[x.dtype], size_b,
                                  step_b, idx, alpha, gain, clamp, _lib.stream_handle())
    _lib.check(rc, "h3d_bias_act")
    return y
