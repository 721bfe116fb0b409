"""The discriminator's resampling / activation glue on two fused HIP kernels (csrc/pool_up.hip), differentiable to any order.

Reference: lib/discriminators/unet_discriminators.py:8-72 -- nn.Sequential(LeakyReLU(0.2), nn.Upsample(scale_factor=2), conv) in the
up blocks, nn.AvgPool2d(2) and the residual sum in ResBlock.forward.  The two kernels

    up2 (x; mask, slope, scale, addend)[b, :, 2y+i, 2x+j] = scale * m(mask[b,:,y,x]) * x[b,:,y,x] (+ addend)
    pool2(x; x2, mask, slope, scale)   [b, :, y, x]       = scale * m(mask[b,:,y,x]) * sum_ij (x (+ x2))[b,:,2y+i,2x+j]

(m = LeakyReLU's derivative: 1 where mask > 0, `slope` elsewhere; 1 without a mask) are adjoint for a fixed mask, so each is the
other's backward and the pair is closed under differentiation -- what the R1 penalty (a gradient of a gradient) needs.  The mask is
treated as a constant: LeakyReLU's second derivative is zero almost everywhere, which is also what torch's autograd uses.

    lrelu_up2(x)      = up2(x, mask = x)            up(lrelu(x)) in one pass, no lrelu(x) tensor
    up2_add(s, d)     = up2(s, addend = d)          up(s) + d
    avgpool2_sum(s,d) = pool2(s, x2 = d, 1/4)       avgpool(s + d), no full-resolution sum
Logical NCHW tensors in channels-last memory (what the native convolutions produce); fp32, or f16 in the AMP tier."""
import torch

from .... import _lib


def supported(*tensors):
    t0 = tensors[0]
    return all(t is None or (t.is_cuda and t.dim() == 4 and t.dtype == t0.dtype) for t in tensors) and \
        t0.dtype in (torch.float32, torch.float16)


def _cl(t):
    """the tensor in dense channels-last memory, 16-byte aligned (a copy only when it is not already)"""
    return None if t is None else _lib.aligned16(t.detach().contiguous(memory_format=torch.channels_last))


def _up2(x, mask, addend, slope, scale):
    x, mask, addend = _cl(x), _cl(mask), _cl(addend)
    B, C, H, W = x.shape
    out = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    rc = _lib.load().h3d_up2_mask(_lib.ptr(x), _lib.ptr(mask), _lib.ptr(addend), _lib.ptr(out), B, H, W, C, float(slope), float(scale),
                                  int(x.dtype == torch.float16), _lib.stream_handle())
    _lib.check(rc, "h3d_up2_mask")
    return out


def _pool2(x, x2, mask, slope, scale):
    x, x2, mask = _cl(x), _cl(x2), _cl(mask)
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise ValueError(f"pool2: odd size {H}x{W}")
    out = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    rc = _lib.load().h3d_pool2_mask(_lib.ptr(x), _lib.ptr(x2), _lib.ptr(mask), _lib.ptr(out), B, H // 2, W // 2, C, float(slope),
                                    float(scale), int(x.dtype == torch.float16), _lib.stream_handle())
    _lib.check(rc, "h3d_pool2_mask")
    return out


class _Up2(torch.autograd.Function):
    """up2(x; mask, slope, scale) (+ addend).  d/dx = pool2(.; mask, slope, scale), d/d(addend) = identity; mask: constant."""

    @staticmethod
    def forward(ctx, x, mask, addend, slope, scale):
        ctx.save_for_backward(mask)
        ctx.slope, ctx.scale, ctx.has_add = slope, scale, addend is not None
        return _up2(x, mask, addend, slope, scale)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        gx = _Pool2.apply(g, None, mask, ctx.slope, ctx.scale) if ctx.needs_input_grad[0] else None
        return gx, None, (g if ctx.has_add and ctx.needs_input_grad[2] else None), None, None


class _Pool2(torch.autograd.Function):
    """pool2(x (+ x2); mask, slope, scale).  d/dx = d/dx2 = up2(.; mask, slope, scale): ONE tensor serves both inputs."""

    @staticmethod
    def forward(ctx, x, x2, mask, slope, scale):
        ctx.save_for_backward(mask)
        ctx.slope, ctx.scale, ctx.two = slope, scale, x2 is not None
        return _pool2(x, x2, mask, slope, scale)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        need2 = ctx.two and ctx.needs_input_grad[1]
        u = _Up2.apply(g, mask, None, ctx.slope, ctx.scale) if (ctx.needs_input_grad[0] or need2) else None
        return (u if ctx.needs_input_grad[0] else None), (u if need2 else None), None, None, None


def lrelu_up2(x, slope=0.2):
    """F.interpolate(F.leaky_relu(x, slope), scale_factor=2, mode="nearest")"""
    return _Up2.apply(x, x.detach(), None, slope, 1.0)


def up2(x):
    """F.interpolate(x, scale_factor=2, mode="nearest")"""
    return _Up2.apply(x, None, None, 1.0, 1.0)


def up2_add(s, d):
    """F.interpolate(s, scale_factor=2, mode="nearest") + d"""
    return _Up2.apply(s, None, d, 1.0, 1.0)


def avgpool2(x):
    """F.avg_pool2d(x, 2)"""
    return _Pool2.apply(x, None, None, 1.0, 0.25)


def avgpool2_sum(s, d):
    """F.avg_pool2d(s + d, 2)"""
    return _Pool2.apply(s, d, None, 1.0, 0.25)
