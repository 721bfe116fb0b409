"""Test infrastructure: a plain-torch stand-in for the four-function kernel set of 3dhumangan_amd.lib.components.ops.spade
(HipKernels).  It is (a) the reference the HIP kernels are compared with on the GPU and (b) what lets the world-2 gloo test
run the collective algebra of spade_norm_act on CPU.  Not part of the product."""
import torch

SLOPE = 0.2


def _bc(t, x):
    return t if t.dim() == 3 else t[:, None, :]


class TorchKernels:
    def moments(self, x):
        xd = x.double()
        return torch.stack([xd.sum((0, 1)), (xd * xd).sum((0, 1))])

    def forward(self, x, scale, shift, gamma, beta):
        u = (x * scale + shift) * (1 + _bc(gamma, x)) + _bc(beta, x)
        return torch.where(u > 0, u, u * SLOPE)

    def _chain(self, x, mean, rstd, g, b, gamma, beta, dy):
        n = (x - mean) * rstd
        h = n * g + b
        u = h * (1 + _bc(gamma, x)) + _bc(beta, x)
        du = torch.where(u > 0, dy, dy * SLOPE)
        return n, h, du, du * (1 + _bc(gamma, x))

    def backward_sums(self, x, mean, rstd, g, b, gamma, beta, dy):
        n, _, _, dh = self._chain(x, mean, rstd, g, b, gamma, beta, dy)
        return torch.stack([dh.double().sum((0, 1)), (dh * n).double().sum((0, 1))])

    def backward_apply(self, x, mean, rstd, g, b, gamma, beta, dy, c1, c2):
        n, h, du, dh = self._chain(x, mean, rstd, g, b, gamma, beta, dy)
        dx = rstd * g * (dh - c1 - n * c2)
        dgamma, dbeta = du * h, du
        if gamma.dim() == 2:
            dgamma, dbeta = dgamma.sum(1), dbeta.sum(1)
        return dx, dgamma, dbeta
