"""upfirdn2d family with the reference's Python signatures (lib/components/ops/upfirdn2d.py:69-161, 276-386),
executed by the HIP kernel behind h3d_upfirdn2d.  Differentiable in x to any order: the adjoint of upfirdn2d is upfirdn2d with
up / down swapped, the filter flipped and the complementary padding (reference upfirdn2d.py:230-264, Upfirdn2dCuda.backward), so
the backward pass runs on the same kernel."""
import ctypes

import numpy as np
import torch

from .... import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def _pair(v):
    if isinstance(v, int):
        return v, v
    v = list(v)
    assert len(v) == 2 and all(isinstance(e, int) for e in v)
    return v[0], v[1]


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    assert all(isinstance(e, int) for e in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return padding


def _filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])      # (fw, fh)


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """2-D (or separable 1-D) low-pass filter constant; reference upfirdn2d.py:69-113."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    B, C, H, W = x.shape
    fh, fw = f2d.shape
    outW = (W * upx + px0 + px1 - fw + downx) // downx
    outH = (H * upy + py0 + py1 - fh + downy) // downy
    if outW < 1 or outH < 1:
        raise RuntimeError("upfirdn2d: output must be at least 1x1")
    y = torch.empty((B, C, outH, outW), device=x.device, dtype=x.dtype,
                    memory_format=torch.channels_last if x.is_contiguous(memory_format=torch.channels_last)
                    and not x.is_contiguous() else torch.contiguous_format)
    xs = (ctypes.c_int64 * 4)(*x.stride())
    ys = (ctypes.c_int64 * 4)(*y.stride())
    rc = _lib.load().h3d_upfirdn2d(_lib.ptr(x), _lib.ptr(f2d), _lib.ptr(y), _DTYPES[x.dtype], B, C, H, W, xs, fh, fw,
                                   outH, outW, ys, upx, upy, downx, downy, px0, py0, int(bool(flip)), float(gain),
                                   _lib.stream_handle())
    _lib.check(rc, "h3d_upfirdn2d")
    return y


def _forward(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain):
    if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        x = x.contiguous()
    if f.ndim == 2:
        return _launch(x, f.contiguous(), upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    # separable: the reference's kernel wrapper runs a [1,n] pass then an [n,1] pass, each with gain**0.5
    g = float(gain) ** 0.5
    y = _launch(x, f.unsqueeze(0).contiguous(), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, g)
    return _launch(y, f.unsqueeze(1).contiguous(), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, g)


class _Upfirdn2d(torch.autograd.Function):
    """x -> upfirdn2d(x); the gradient w.r.t. x is upfirdn2d of dy with up / down swapped, the filter flipped and the
    complementary padding (reference Upfirdn2dCuda.backward, upfirdn2d.py:248-262) -- itself differentiable."""

    @staticmethod
    def forward(ctx, x, f, params):
        ctx.params, ctx.x_shape = params, x.shape
        ctx.save_for_backward(f)
        return _forward(x, f, *params)

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain = ctx.params
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = dy.shape
        fw, fh = _filter_size(f)
        p = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = upfirdn2d(dy, f, up=(downx, downy), down=(upx, upy), padding=p, flip_filter=not flip_filter, gain=gain)
        return dx, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="hip"):
    """Pad, upsample, filter, and downsample a batch of 2D images (reference upfirdn2d.py:117-161).
    ``impl`` is accepted for compatibility; everything runs on the HIP kernel."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if x.dtype not in _DTYPES:
        raise TypeError(f"upfirdn2d: unsupported dtype {x.dtype}")
    _lib.need_cuda(x, f)
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert f.dtype == torch.float32 and f.ndim in (1, 2)
    f = f.to(x.device)
    params = (upx, upy, downx, downy, px0, px1, py0, py1, bool(flip_filter), gain)
    if x.requires_grad and torch.is_grad_enabled():
        return _Upfirdn2d.apply(x, f, params)
    return _forward(x.detach(), f, *params)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="hip"):
    """reference upfirdn2d.py:276-310."""
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="hip"):
    """reference upfirdn2d.py:314-349."""
    upx, upy = _pair(up)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="hip"):
    """reference upfirdn2d.py:353-386."""
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
