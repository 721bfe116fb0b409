"""BatchNorm + SPADE modulation + LeakyReLU of a SPADEBlock half (reference lib/components/map3d_layers.py:176-190, 228-233)
as one autograd node over channels-last activations, executed by the HIP kernels of csrc/spade_train.hip:

    y = lrelu_0.2( ((x - mean) * rstd * g + b) * (1 + gamma) + beta )

forward = one pass (+ one moments pass in train mode), backward = a reduction pass and an apply pass; the batch-statistics
terms of the BatchNorm backward are applied analytically (no graph through mean / var), and with a process group the two
[2, C] moment vectors are all-reduced -- nn.SyncBatchNorm's exchange (the reference's first_norm, map3d_layers.py:162).

The arithmetic lives behind a four-function kernel set (`HipKernels`); the collective algebra around it is independent of it,
which is what the world-2 gloo test exercises with a stand-in kernel set of its own (there is no CPU kernel set in the
product)."""
import torch
import torch.distributed as dist

from .... import _lib

SLOPE = 0.2


class HipKernels:
    """x [B,P,C] contiguous, fp32 or (AMP tier) f16; gamma / beta [B,P,C] (per pixel, same type as x) or [B,C] (per sample, fp32);
    per-channel vectors [C] fp32.  f16 tensors go to the _f16 entry points: same passes, fp32 arithmetic in registers."""

    @staticmethod
    def _fn(name, x):
        return getattr(_lib.load(), name + ("_f16" if x.dtype == torch.float16 else ""))

    @staticmethod
    def _nblk(P):
        rows = _lib.load().h3d_spade_rows()
        return (P + rows - 1) // rows

    def moments(self, x):
        """-> [2, C] float64: sum x, sum x^2 over all rows."""
        B, P, C = x.shape
        partial = torch.empty((B, self._nblk(P), 2, C), device=x.device, dtype=torch.float32)
        _lib.check(self._fn("h3d_channel_moments", x)(_lib.ptr(x), _lib.ptr(partial), B, P, C, _lib.stream_handle()),
                   "h3d_channel_moments")
        return partial.double().sum(dim=(0, 1))

    def forward(self, x, scale, shift, gamma, beta):
        B, P, C = x.shape
        y = torch.empty_like(x)
        _lib.check(self._fn("h3d_spade_fwd", x)(_lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(gamma), _lib.ptr(beta),
                                             _lib.ptr(y), B, P, C, int(gamma.dim() == 3), SLOPE, _lib.stream_handle()),
                   "h3d_spade_fwd")
        return y

    def backward_sums(self, x, mean, rstd, g, b, gamma, beta, dy):
        """-> [2, C] float64: sum dh, sum dh * n."""
        B, P, C = x.shape
        partial = torch.empty((B, self._nblk(P), 2, C), device=x.device, dtype=torch.float32)
        _lib.check(self._fn("h3d_spade_bwd_reduce", x)(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(g), _lib.ptr(b),
                                                    _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(dy), _lib.ptr(partial), B, P, C,
                                                    int(gamma.dim() == 3), SLOPE, _lib.stream_handle()),
                   "h3d_spade_bwd_reduce")
        return partial.double().sum(dim=(0, 1))

    def backward_apply(self, x, mean, rstd, g, b, gamma, beta, dy, c1, c2, add1=None, add2=None):
        """-> dx, dgamma, dbeta (shaped like gamma / beta).  add1 / add2 (x's shape and type): further gradients of x, added into dx
        by the same pass (h3d_spade_bwd_apply_acc)."""
        B, P, C = x.shape
        pix = gamma.dim() == 3
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(x) if pix else None
        dbeta = torch.empty_like(x) if pix else None
        partial = None if pix else torch.empty((B, self._nblk(P), 2, C), device=x.device, dtype=torch.float32)
        if add1 is not None or add2 is not None:
            rc = _lib.load().h3d_spade_bwd_apply_acc(int(x.dtype == torch.float16), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(g),
                                                     _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(dy), _lib.ptr(c1), _lib.ptr(c2),
                                                     _lib.ptr(add1), _lib.ptr(add2), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                                     _lib.ptr(partial), B, P, C, int(pix), SLOPE, _lib.stream_handle())
            _lib.check(rc, "h3d_spade_bwd_apply_acc")
        else:
            _lib.check(self._fn("h3d_spade_bwd_apply", x)(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(g), _lib.ptr(b),
                                                       _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(dy), _lib.ptr(c1), _lib.ptr(c2),
                                                       _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(partial), B, P, C,
                                                       int(pix), SLOPE, _lib.stream_handle()), "h3d_spade_bwd_apply")
        if not pix:
            sums = partial.sum(dim=1)
            dgamma, dbeta = sums[:, 0], sums[:, 1]
        return dx, dgamma, dbeta


_HIP = HipKernels()


def _sync_on(group):
    """group: a process group, None (= the default group when torch.distributed is initialised) or False (never synchronise)."""
    return group is not False and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class _SpadeNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, gamma, beta, mean, rstd, count, group, kernels, aliases=0):
        # count: global number of rows behind mean / rstd when they are batch statistics (device scalar), None for running ones
        # aliases (round 6): the node also hands out that many views of x.  What reads x through them (the residual connection, a
        # ToRGB head) sends its gradient HERE, where the backward pass adds it into dx as it writes it -- instead of autograd
        # summing the gradients of x's consumers in passes of its own.
        scale = (rstd * g).contiguous()
        shift = (b - mean * scale).contiguous()
        ctx.save_for_backward(x, g, b, gamma, beta, mean, rstd, count)
        ctx.group, ctx.kernels = group, kernels
        y = kernels.forward(x, scale, shift, gamma, beta)
        return y if not aliases else (y,) + tuple(x.view_as(x) for _ in range(aliases))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *d_alias):
        x, g, b, gamma, beta, mean, rstd, count = ctx.saved_tensors
        k = ctx.kernels
        dy = dy.contiguous().to(x.dtype)
        sums = k.backward_sums(x, mean, rstd, g, b, gamma, beta, dy)              # local: they are d_b, d_g
        d_b, d_g = sums[0].float(), sums[1].float()
        if count is not None:
            if _sync_on(ctx.group):
                sums = sums.clone()
                dist.all_reduce(sums, group=ctx.group)
            c = (sums / count.double()).float()
            c1, c2 = c[0].contiguous(), c[1].contiguous()
        else:
            c1 = c2 = torch.zeros_like(mean)
        extra = [d.contiguous().to(x.dtype) for d in d_alias if d is not None]
        if extra and isinstance(k, HipKernels) and len(extra) <= 2:
            dx, dgamma, dbeta = k.backward_apply(x, mean, rstd, g, b, gamma, beta, dy, c1, c2, *extra)
        else:
            dx, dgamma, dbeta = k.backward_apply(x, mean, rstd, g, b, gamma, beta, dy, c1, c2)
            for d in extra:
                dx = dx + d
        if gamma.dim() == 2:
            dgamma, dbeta = dgamma.contiguous(), dbeta.contiguous()
        return dx, d_g, d_b, dgamma, dbeta, None, None, None, None, None, None


def spade_norm_act(x, norm, gamma, beta, training, group=None, eps=1e-5, momentum=0.1, kernels=None, aliases=0):
    """x [B,P,C]; norm: the first_norm parameter holder (weight, bias, running_mean, running_var, num_batches_tracked);
    gamma / beta [B,P,C] or [B,1,C].  training: batch statistics (all-reduced over `group`) + running-statistics update.
    aliases > 0: -> (y, x_1, .., x_aliases), views of x whose gradients are added into dx by the backward kernel itself."""
    k = _HIP if kernels is None else kernels
    if k is _HIP:
        _lib.need_cuda(x, gamma, beta)
    B, P, C = x.shape
    # AMP tier: f16 activations stay f16 through the kernels (the per-pixel gamma / beta travel in the same type); anything else
    # (bf16 autocast, float64) is brought to fp32
    dt = torch.float16 if (x.dtype == torch.float16 and k is _HIP) else torch.float32
    x = x.contiguous().to(dt)
    if gamma.shape[1] == 1 and P != 1:
        gamma, beta = gamma.reshape(B, C).float(), beta.reshape(B, C).float()
    else:
        gamma, beta = gamma.to(dt), beta.to(dt)
    gamma, beta = gamma.contiguous(), beta.contiguous()
    count = None
    if training:
        with torch.no_grad():
            sums = k.moments(x)
            count = torch.tensor([float(B * P)], device=x.device, dtype=torch.float64)
            if _sync_on(group):
                packed = torch.cat([sums.flatten(), count])
                dist.all_reduce(packed, group=group)
                sums, count = packed[:-1].reshape(2, C), packed[-1:]
            mean64 = sums[0] / count
            var64 = (sums[1] / count - mean64 * mean64).clamp_min(0)
            mean, var = mean64.float(), var64.float()
            norm.running_mean.lerp_(mean.to(norm.running_mean.dtype), momentum)
            norm.running_var.lerp_((var64 * (count / (count - 1).clamp_min(1))).to(norm.running_var.dtype), momentum)
            norm.num_batches_tracked += 1
    else:
        mean, var = norm.running_mean.float(), norm.running_var.float()
    rstd = torch.rsqrt(var + eps)
    return _SpadeNormAct.apply(x, norm.weight.float(), norm.bias.float(), gamma, beta, mean.contiguous(), rstd.contiguous(),
                               count, group, k, int(aliases))
