"""Modulated per-pixel layers of the reference's lib/components/map3d_layers.py that are named by the north_star
(the SPADE path of the shipped configs lives in lib/generators/synthesis_pack.py + csrc/synthesis*.hip).

SpatialStyleModLayer: per-pixel modulated 1x1 convolution with demodulation (reference :25-80), evaluated by
h3d_modconv1x1 on the fp32 matrix cores.  Same constructor, parameter names/shapes and forward signature."""
import math

import torch
import torch.nn as nn

from ... import _lib
from ..generators.synthesis_pack import pack_matrix
from .ops.linear import linear as native_linear


def _pad_vec(v, n):
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.flatten().float()
    return out


class SpatialStyleModLayer(nn.Module):

    def __init__(self, in_channel, out_channel, kernel_size=1, style_dim=None, demodulate=True, eps=1e-8, **kwargs):
        super().__init__()
        assert kernel_size == 1
        self.eps, self.in_channel, self.out_channel = eps, in_channel, out_channel
        self.kernel_size, self.style_dim, self.demodulate = kernel_size, style_dim, demodulate
        self.weight = nn.Parameter(torch.randn(1, 1, in_channel, out_channel) * math.sqrt(2 / (1 + 0.2 ** 2)) / math.sqrt(in_channel))
        self.bias = nn.Parameter(torch.zeros(1, 1, out_channel))
        self.affine = nn.Linear(style_dim, in_channel)
        nn.init.kaiming_normal_(self.affine.weight, mode="fan_in", nonlinearity="linear")
        self._packed = None

    def _pack(self, device):
        ps = (self.weight, self.bias, self.affine.weight, self.affine.bias)
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or self._packed[0] != key:
            Cin, Cout, S = self.in_channel, self.out_channel, self.style_dim
            r32 = lambda n: (n + 31) // 32 * 32
            w = self.weight.detach()[0, 0].to(device).float()                  # [Cin, Cout]
            wa = self.affine.weight.detach().to(device).float()                # [Cin, S]
            self._packed = (key, dict(
                w_aff=pack_matrix(wa, r32(S) // 8, r32(Cin) // 32),
                b_aff=_pad_vec(self.affine.bias.detach().to(device) + 1.0, r32(Cin)),
                w=pack_matrix(w.t().contiguous(), r32(Cin) // 8, r32(Cout) // 32),
                w2=pack_matrix((w * w).t().contiguous(), r32(Cin) // 8, r32(Cout) // 32),
                bias=_pad_vec(self.bias.detach().to(device), r32(Cout))))
        return self._packed[1]

    def forward(self, x, style):
        """x [B,P,Cin]; style [B,P,S] or [B,S,H,W]  ->  [B,P,Cout].
        Without gradients: one fused kernel (h3d_modconv1x1, fp32 matrix cores).  With gradients (round 4): the same function,
        ((x m) W) d + b with m = affine(style) + 1, d = rsqrt(m^2 W^2 + eps) (reference lib/components/map3d_layers.py:60-80),
        composed of the package's own dense layers (ops/linear.py: forward / data-gradient GEMMs on h3d_conv_x3 where the
        widths allow, weight gradients on h3d_wgrad_x3 / h3d_wgrad_narrow) -- recorded by autograd.  FIRST order only: the dense
        layers' backward functions (ops/linear.py: _LinearX3, _LinearAmp) are once_differentiable, so a double backward through this
        layer raises; the k x k StyleModLayer (cips_layers.py) goes through ops/conv.py's three mutually recursive primitives and is
        differentiable to any order."""
        _lib.need_cuda(x, style)
        if style.dim() > 3:
            B, C, H, W = style.shape
            style = style.permute(0, 2, 3, 1).reshape(B, H * W, C)
        tensors = (x, style, self.weight, self.bias, self.affine.weight, self.affine.bias)
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            return self._composed(*tensors)
        with torch.no_grad():
            return self._launch(x, style)

    def _composed(self, x, style, weight, bias, aw, ab):
        m = native_linear(style.float(), aw, ab) + 1.0
        wt = weight[0, 0].t()                                    # [Cout, Cin]: the layout of a dense layer's weight
        y = native_linear(x.float() * m, wt)
        if self.demodulate:
            y = y * torch.rsqrt(native_linear(m * m, wt * wt) + self.eps)
        return y + bias[0]

    def _launch(self, x, style, *_):
        B, P, Cin = x.shape
        pk = self._pack(x.device)
        xin = x.contiguous().float()
        st = style.contiguous().float()
        out = torch.empty(B, P, self.out_channel, device=x.device, dtype=torch.float32)
        rc = _lib.load().h3d_modconv1x1(_lib.ptr(xin), _lib.ptr(st), _lib.ptr(pk["w_aff"]), _lib.ptr(pk["b_aff"]),
                                        _lib.ptr(pk["w"]), _lib.ptr(pk["w2"]), _lib.ptr(pk["bias"]), _lib.ptr(out),
                                        B * P, Cin, self.out_channel, self.style_dim, int(bool(self.demodulate)),
                                        float(self.eps), _lib.stream_handle())
        _lib.check(rc, "h3d_modconv1x1")
        return out
