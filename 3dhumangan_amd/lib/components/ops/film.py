"""sin(freq * x + phase): the activation of the reference's SineLayer / FiLMLayer (lib/components/pigan_layers.py:63-87) as one
fused HIP pass each way (h3d_film_sin / h3d_film_sin_bwd).  The backward recomputes the cosine from the saved Linear output,
so a layer keeps ONE activation-sized tensor alive instead of the three torch's autograd would."""
import torch

from .... import _lib

_DT = {torch.float32: 0, torch.float16: 1}


def _run_fwd(x, freq, phase, w0):
    B, N, C = x.shape
    y = torch.empty_like(x)
    rc = _lib.load().h3d_film_sin(_lib.ptr(x), _lib.ptr(freq), _lib.ptr(phase), _lib.ptr(y), B, N, C, _DT[x.dtype], float(w0),
                                  _lib.stream_handle())
    _lib.check(rc, "h3d_film_sin")
    return y


class _FilmSin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, freq, phase, w0):
        ctx.save_for_backward(x, freq, phase)
        ctx.w0 = w0
        return _run_fwd(x, freq, phase, w0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, freq, phase = ctx.saved_tensors
        B, N, C = x.shape
        lib = _lib.load()
        dy = dy.contiguous().to(x.dtype)
        dx = torch.empty_like(x)
        want_fp = freq is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        partial = None
        if want_fp:
            nblk = (N + lib.h3d_film_sin_rows() - 1) // lib.h3d_film_sin_rows()
            partial = torch.empty((B, nblk, 2, C), device=x.device, dtype=torch.float32)
        rc = lib.h3d_film_sin_bwd(_lib.ptr(x), _lib.ptr(freq), _lib.ptr(phase), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(partial),
                                  B, N, C, _DT[x.dtype], float(ctx.w0), _lib.stream_handle())
        _lib.check(rc, "h3d_film_sin_bwd")
        d_freq = d_phase = None
        if want_fp:
            sums = partial.sum(dim=1)
            d_freq, d_phase = sums[:, 0], sums[:, 1]
        return (dx if ctx.needs_input_grad[0] else None), d_freq, d_phase, None


def film_sin(x, freq=None, phase=None, w0=1.0):
    """x [B,N,C] (fp32 or fp16), freq / phase [B,C] -> sin(freq * x + phase); without freq / phase: sin(w0 * x)."""
    _lib.need_cuda(x, freq, phase)
    if x.dtype == torch.bfloat16:                     # bf16 autocast: the kernels take f32 / f16 activations
        x = x.float()
    if x.dtype not in _DT:
        raise TypeError(f"film_sin: unsupported dtype {x.dtype}")
    assert x.ndim == 3 and (freq is None) == (phase is None)
    x = x.contiguous()
    if freq is not None:
        assert freq.shape == phase.shape == (x.shape[0], x.shape[2])
        freq, phase = freq.contiguous().float(), phase.contiguous().float()
    needs = torch.is_grad_enabled() and (x.requires_grad or (freq is not None and (freq.requires_grad or phase.requires_grad)))
    if needs:
        return _FilmSin.apply(x, freq, phase, w0)
    return _run_fwd(x, freq, phase, w0)
