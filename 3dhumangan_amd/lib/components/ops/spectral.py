"""Spectral normalisation of a convolution weight on fused HIP kernels (csrc/spectral_norm.hip).

Reference: torch.nn.utils.spectral_norm, which the discriminator wraps every convolution in
(lib/discriminators/unet_discriminators.py:17); this is SpectralNorm.compute_weight with n_power_iterations = 1 in training mode:
    v <- normalize(W^T u),  u <- normalize(W v)   (buffers, in place, no gradient)
    W_sn = W / (u . W v)                          (gradient through W with u, v constant)
Three launches forward and two backward instead of torch's ~13 + ~8 per layer and pass."""
import torch

from .... import _lib


class _SpectralWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, u_buf, v_buf, eps):
        lib = _lib.load()
        R = w.shape[0]
        K = w.numel() // R
        wc = _lib.aligned16(w.detach().contiguous())
        dev = w.device
        u_out = torch.empty(R, device=dev, dtype=torch.float32)
        v_out = torch.empty(K, device=dev, dtype=torch.float32)
        sigma = torch.empty(1, device=dev, dtype=torch.float32)
        w_sn = torch.empty_like(wc)
        scratch = torch.empty(lib.h3d_spectral_norm_scratch(R, K), device=dev, dtype=torch.float32)
        rc = lib.h3d_spectral_norm(_lib.ptr(wc), _lib.ptr(u_buf), _lib.ptr(u_out), _lib.ptr(u_buf), _lib.ptr(v_out), _lib.ptr(v_buf),
                                   _lib.ptr(sigma), _lib.ptr(w_sn), _lib.ptr(scratch), R, K, float(eps), _lib.stream_handle())
        _lib.check(rc, "h3d_spectral_norm")
        ctx.save_for_backward(w_sn, u_out, v_out, sigma)      # this call's u', v' (the buffers move on with the next forward)
        return w_sn

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w_sn, u, v, sigma = ctx.saved_tensors
        lib = _lib.load()
        R = w_sn.shape[0]
        K = w_sn.numel() // R
        g = g.contiguous().float()
        dw = torch.empty_like(w_sn)
        scratch = torch.empty(lib.h3d_spectral_norm_scratch(R, K), device=g.device, dtype=torch.float32)
        rc = lib.h3d_spectral_norm_bwd(_lib.ptr(g), _lib.ptr(w_sn), _lib.ptr(u), _lib.ptr(v), _lib.ptr(sigma), _lib.ptr(dw),
                                       _lib.ptr(scratch), R, K, _lib.stream_handle())
        _lib.check(rc, "h3d_spectral_norm_bwd")
        return dw, None, None, None


def supported(module, hook):
    """The fused path covers what the discriminator uses: training mode (one power iteration per forward), dim 0, fp32 CUDA
    parameters, buffers that are dense."""
    w = getattr(module, hook.name + "_orig", None)
    u, v = getattr(module, hook.name + "_u", None), getattr(module, hook.name + "_v", None)
    return (module.training and hook.n_power_iterations == 1 and hook.dim == 0 and w is not None and w.is_cuda
            and w.dtype == torch.float32 and u is not None and v is not None and u.is_contiguous() and v.is_contiguous()
            and u.dtype == torch.float32 and v.dtype == torch.float32)


def spectral_weight(module, hook):
    """module.<name> for this forward: W_orig / sigma after one power iteration (u, v buffers updated in place)."""
    w = getattr(module, hook.name + "_orig")
    return _SpectralWeight.apply(w, getattr(module, hook.name + "_u"), getattr(module, hook.name + "_v"), hook.eps)
