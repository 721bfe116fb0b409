"""StyleModLayer -- the StyleGAN2 modulated convolution of the reference's lib/components/cips_layers.py:155-294
(the "modulated conv2d" the north_star names; no shipped config instantiates it).  forward_group_conv is
evaluated by h3d_modconv2d: the per-sample grouped convolution is rewritten, exactly, as "modulate the input,
shared-weight implicit GEMM on the fp32 matrix cores, demodulate the output"."""
import torch
import torch.nn as nn

from ... import _lib
from ..generators.synthesis_pack import pack_matrix
from .ops import conv as native_conv


def _pad_rows(v, n):
    out = torch.zeros(v.shape[0], n, dtype=torch.float32, device=v.device)
    out[:, : v.shape[1]] = v.float()
    return out


class StyleModLayer(nn.Module):

    def __init__(self, in_channel, out_channel, kernel_size=1, style_dim=None, demodulate=True, use_group_conv=True,
                 eps=1e-8, **kwargs):
        super().__init__()
        self.eps, self.in_channel, self.out_channel = eps, in_channel, out_channel
        self.kernel_size, self.style_dim, self.demodulate = kernel_size, style_dim, demodulate
        self.use_group_conv = use_group_conv
        self.padding = kernel_size // 2
        if use_group_conv:
            self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        else:
            assert kernel_size == 1
            self.weight = nn.Parameter(torch.randn(1, in_channel, out_channel))
        nn.init.kaiming_normal_(self.weight[0], a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        self.bias = nn.Parameter(torch.zeros(1, out_channel))
        self.geo_feature = nn.Linear(style_dim, in_channel)
        nn.init.kaiming_normal_(self.geo_feature.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        self._packed = None

    def _weight_oikk(self):
        """[Cout, Cin, k, k] view of either parameterisation."""
        if self.use_group_conv:
            return self.weight[0]
        return self.weight[0].t().reshape(self.out_channel, self.in_channel, 1, 1)

    def _pack(self, device):
        key = (str(device), self.weight.data_ptr(), self.weight._version, self.bias._version)
        if self._packed is None or self._packed[0] != key:
            w = self._weight_oikk().detach().to(device).float()
            Cout, Cin, k, _ = w.shape
            r32 = lambda n: (n + 31) // 32 * 32
            taps = [pack_matrix(w[:, :, ky, kx].contiguous(), r32(Cin) // 8, r32(Cout) // 32)
                    for ky in range(k) for kx in range(k)]
            bias = torch.zeros(r32(Cout), device=device)
            bias[:Cout] = self.bias.detach().to(device).flatten()
            self._packed = (key, dict(w=torch.cat(taps).contiguous(), w2sum=(w * w).sum(dim=(2, 3)), bias=bias))
        return self._packed[1]

    def forward(self, x, style):
        """x [B,Cin,H,W] | [B,Cin] | [B,N,Cin]; style [B,S]  ->  same layout with Cout channels.
        Without gradients: one fused kernel (h3d_modconv2d, fp32 matrix cores).  With gradients (round 4): the same function
        composed of native primitives -- modulate the input, the shared-weight convolution of csrc/conv_x3.hip (split bf16 on the
        matrix cores; its three autograd primitives are closed under differentiation, lib/components/ops/conv.py), demodulate --
        so forward, data gradient and weight gradient all run on hand-written kernels and no library convolution appears in
        the graph (round 3 recomputed a tensor-algebra restatement on library convolutions in backward)."""
        _lib.need_cuda(x, style)
        assert x.shape[0] == style.shape[0]
        if x.dim() not in (2, 3, 4):
            raise Exception("wrong input size")
        tensors = (x, style, self.weight, self.bias, self.geo_feature.weight, self.geo_feature.bias)
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            return self._composed(*tensors)
        with torch.no_grad():
            return self._launch(x, style)

    @staticmethod
    def _to_image(x):
        if x.dim() == 2:
            return x[:, :, None, None]
        if x.dim() == 3:
            return x.permute(0, 2, 1).unsqueeze(-1)
        return x

    @staticmethod
    def _from_image(out, dim):
        if dim == 2:
            return out[:, :, 0, 0]
        if dim == 3:
            return out[:, :, :, 0].permute(0, 2, 1).contiguous()
        return out

    def _oikk(self, weight):
        if self.use_group_conv:
            return weight[0]
        return weight[0].t().reshape(self.out_channel, self.in_channel, 1, 1)

    def _composed(self, x, style, weight, bias, gw, gb):
        """The grouped convolution (reference lib/components/cips_layers.py:235-278) as: modulate the input, shared-weight
        convolution, demodulate the output -- exact algebra, the convolution on the native kernels (k in {1, 3}; any other
        kernel size goes to the library convolution)."""
        w = self._oikk(weight).float()
        s = torch.nn.functional.linear(style.float(), gw, gb) + 1.0
        xm = self._to_image(x).float() * s[:, :, None, None]
        if native_conv.supported(xm, w):
            y = native_conv.conv2d(xm.contiguous(memory_format=torch.channels_last), w)
        else:
            y = torch.nn.functional.conv2d(xm, w, padding=self.padding)
        if self.demodulate:
            y = y * torch.rsqrt((s * s) @ (w * w).sum(dim=(2, 3)).t() + self.eps)[:, :, None, None]
        return self._from_image(y + bias.view(1, -1, 1, 1), x.dim())

    def _launch(self, x, style, *_):
        inp = self._to_image(x)
        B, Cin, H, W = inp.shape
        pk = self._pack(x.device)
        r32 = lambda n: (n + 31) // 32 * 32
        s = self.geo_feature(style.float()) + 1.0                                     # [B, Cin]  (tiny library GEMM)
        if self.demodulate:
            d = torch.rsqrt((s * s) @ pk["w2sum"].t() + self.eps)                    # [B, Cout]
        else:
            d = torch.ones(B, self.out_channel, device=x.device)
        smod, dmod = _pad_rows(s, r32(Cin)), _pad_rows(d, r32(self.out_channel))
        xin = inp.contiguous().float()
        out = torch.empty(B, self.out_channel, H, W, device=x.device, dtype=torch.float32)
        rc = _lib.load().h3d_modconv2d(_lib.ptr(xin), _lib.ptr(smod), _lib.ptr(dmod), _lib.ptr(pk["w"]), _lib.ptr(pk["bias"]),
                                       _lib.ptr(out), B, Cin, self.out_channel, H, W, self.kernel_size, _lib.stream_handle())
        _lib.check(rc, "h3d_modconv2d")
        return self._from_image(out, x.dim())
