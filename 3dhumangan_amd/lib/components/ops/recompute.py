"""Gradients for forward-only HIP kernels by recomputation: the forward value comes from the kernel, the backward re-evaluates a
differentiable restatement of the same function on the device (plain tensor algebra: library GEMMs / convolutions) and pulls the
cotangent through it.  Used by the two modulated-convolution layers no shipped config trains (SpatialStyleModLayer,
StyleModLayer): they get correct gradients for every input and parameter without a dedicated backward kernel; the restatement
runs only in backward, never in a forward."""
import torch


class _Recompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, run, restate, *tensors):
        ctx.restate = restate
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            return run(*tensors)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        need = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(n) for t, n in zip(ctx.saved_tensors, need)]
            y = ctx.restate(*leaves)
            wanted = [l for l in leaves if l.requires_grad]
            got = iter(torch.autograd.grad(y, wanted, dy.to(y.dtype), allow_unused=True))
        return (None, None) + tuple(next(got) if n else None for n in need)


def with_recomputed_grad(run, restate, *tensors):
    """run(*tensors) (a kernel launch; evaluated without recording); differentiable through restate(*tensors)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _Recompute.apply(run, restate, *tensors)
    with torch.no_grad():
        return run(*tensors)
