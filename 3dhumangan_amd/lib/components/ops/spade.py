"""BatchNorm + SPADE modulation + LeakyReLU of a SPADEBlock half (reference lib/components/map3d_layers.py:176-190, 228-233)
as one autograd node over channels-last activations, executed by the HIP kernels of csrc/spade_train.hip:

    y = lrelu_0.2( ((x - mean) * rstd * g + b) * (1 + gamma) + beta )

forward = one pass (+ one moments pass in train mode), backward = a reduction pass and an apply pass; the batch-statistics
terms of the BatchNorm backward are applied analytically (no graph through mean / var), and with a process group the two
[2, C] moment vectors are all-reduced -- nn.SyncBatchNorm's exchange (the reference's first_norm, map3d_layers.py:162).

The arithmetic lives behind a four-function kernel set (`HipKernels`); the collective algebra around it is independent of it,
which is what the world-2 gloo test exercises with a stand-in kernel set of its own (there is no CPU kernel set in the
product)."""
import os

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

    @staticmethod
    def _sum_rows(partial):
        """partial [B, nblk, 2, C] fp32 -> [2, C] float64, one launch (h3d_rows_sum_f64; was a double copy + a reduction)."""
        if not FUSED_BOOKKEEPING:
            return partial.double().sum(dim=(0, 1))
        B, nb, _, C = partial.shape
        out = torch.empty((2, C), device=partial.device, dtype=torch.float64)
        _lib.check(_lib.load().h3d_rows_sum_f64(_lib.ptr(partial), _lib.ptr(out), B * nb, 2 * C, _lib.stream_handle()), "h3d_rows_sum_f64")
        return out

    def moments(self, x):
        """-> [2, C] float64: sum x, sum x^2 over all rows."""
        B, P, C = x.shape
        partial = torch.empty((B, self._nblk(P), 2, C), device=x.device, dtype=torch.float32)
        _lib.check(self._fn("h3d_channel_moments", x)(_lib.ptr(x), _lib.ptr(partial), B, P, C, _lib.stream_handle()),
                   "h3d_channel_moments")
        return self._sum_rows(partial)

    @staticmethod
    def finish(sums, count, norm, eps, momentum):
        """sums [2, C] f64, count [1] f64 (device) -> mean, rstd [C] fp32 + the running-statistics update of `norm`, one launch
        (h3d_bn_finish; was ~15 tensor operations on [C] vectors)."""
        C = sums.shape[1]
        mean = torch.empty(C, device=sums.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        rc = _lib.load().h3d_bn_finish(_lib.ptr(sums), _lib.ptr(count), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(norm.running_mean),
                                       _lib.ptr(norm.running_var), _lib.ptr(norm.num_batches_tracked), C, float(eps), float(momentum),
                                       _lib.stream_handle())
        _lib.check(rc, "h3d_bn_finish")
        return mean, rstd

    @staticmethod
    def backward_finish(local, glob, count):
        """local / glob [2, C] f64 (this rank's / the all-reduced backward sums), count [1] f64 or None -> d_bias, d_weight, c1, c2
        [C] fp32, one launch (h3d_bn_bwd_finish)."""
        C = local.shape[1]
        out = torch.empty((4, C), device=local.device, dtype=torch.float32)
        rc = _lib.load().h3d_bn_bwd_finish(_lib.ptr(local), _lib.ptr(glob), _lib.ptr(count), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                           _lib.ptr(out[2]), _lib.ptr(out[3]), C, _lib.stream_handle())
        _lib.check(rc, "h3d_bn_bwd_finish")
        return out[0], out[1], out[2], out[3]

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
        return self._sum_rows(partial)

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
FUSED_BOOKKEEPING = os.environ.get("H3D_SPADE_BOOKKEEPING", "fused") != "torch"      # round 6: h3d_rows_sum_f64 / h3d_bn_finish / h3d_bn_bwd_finish
_row_counts = {}


def _row_count(n, device):
    """[float(n)] as a float64 device tensor, made once per (n, device): a host-to-device copy per SPADE call otherwise."""
    key = (int(n), str(device))
    t = _row_counts.get(key)
    if t is None:
        t = _row_counts[key] = torch.tensor([float(n)], device=device, dtype=torch.float64)
    return t


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
        if isinstance(k, HipKernels) and FUSED_BOOKKEEPING:
            glob = sums
            if count is not None and _sync_on(ctx.group):
                glob = sums.clone()
                dist.all_reduce(glob, group=ctx.group)
            d_b, d_g, c1, c2 = k.backward_finish(sums, glob if count is not None else None, count)
        else:
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


def spade_norm_act(x, norm, gamma, beta, training, group=None, eps=1e-5, momentum=0.1, kernels=None, aliases=0, moments=None):
    """x [B,P,C]; norm: the first_norm parameter holder (weight, bias, running_mean, running_var, num_batches_tracked);
    gamma / beta [B,P,C] or [B,1,C].  training: batch statistics (all-reduced over `group`) + running-statistics update.
    aliases > 0: -> (y, x_1, .., x_aliases), views of x whose gradients are added into dx by the backward kernel itself.
    moments: [rows, 2, C] fp32 partial column sums of x and x^2 that the layer which produced x took from its accumulators
    (ops/linear.py: linear(.., moments=True)); with them the training-mode pass over x for the batch statistics is skipped."""
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
    fused = (training and isinstance(k, HipKernels) and FUSED_BOOKKEEPING and norm.running_mean is not None
             and norm.running_mean.dtype == torch.float32 and norm.running_var.dtype == torch.float32
             and norm.running_mean.is_contiguous() and norm.running_var.is_contiguous())
    if fused:
        with torch.no_grad():
            if moments is not None and moments.dim() == 3 and moments.shape[1:] == (2, C) and moments.dtype == torch.float32:
                sums = k._sum_rows(moments.unsqueeze(0))
            else:
                sums = k.moments(x)
            count = _row_count(B * P, x.device)
            if _sync_on(group):
                packed = torch.cat([sums.flatten(), count])
                dist.all_reduce(packed, group=group)
                sums, count = packed[:-1].reshape(2, C).contiguous(), packed[-1:].contiguous()
            mean, rstd = k.finish(sums, count, norm, eps, momentum)
        return _SpadeNormAct.apply(x, norm.weight.float(), norm.bias.float(), gamma, beta, mean, rstd, count, group, k, int(aliases))
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
