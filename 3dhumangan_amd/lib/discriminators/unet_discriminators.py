"""UNetDiscriminator -- counterpart of the reference's lib/discriminators/unet_discriminators.py:82-160 (ResBlock :8-72).

Same constructor kwargs (the whole config dict is splatted in), same state_dict schema (spectral-norm convs store
bias / weight_orig / weight_u / weight_v under the reference's module paths, so `*_discriminator` checkpoints load with
strict=True) and the same output dict.  The 3x3 / 1x1 convolutions with channel counts that are multiples of 64 -- all of the
network's arithmetic but the RGB stem, the three heads and the latent layer -- run on the hand-written matrix-core kernels
of csrc/conv_x3.hip / wgrad_x3.hip through lib/components/ops/conv.py (forward, backward-data, weight gradient, and the double
backward of the R1 penalty as compositions of the same three kernels); the remaining small convolutions and any CPU / autocast
call go through torch (MIOpen).  `H3D_DISC_CONV=torch` forces the library path everywhere (A/B measurements).

Module layout note: the reference wraps its convs in nn.Sequential(LeakyReLU, [Upsample,] conv), which fixes the state_dict
index of the conv (conv1.1 / conv1.2 / conv2.1).  Here the activations and resampling are plain functional calls and a
`_Slot` holds the conv under that same index, so only parameters live in modules.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..components.ops import conv as conv_ops
from ..components.ops import pool_up
from ..components.ops import spectral as spectral_ops


class _Slot(nn.Module):
    """Holds one sub-module under a numeric name (`<parent>.<index>.*` state_dict keys) and forwards to it."""

    def __init__(self, index, module):
        super().__init__()
        self.index = str(index)
        self.add_module(self.index, module)

    def forward(self, x):
        return getattr(self, self.index)(x)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (stride 1, padding k // 2) that runs on the native kernels when they cover the call; same parameters, same
    state_dict, works under nn.utils.spectral_norm (the hook sets `weight` before forward)."""

    def forward(self, x):
        hook = getattr(self, "_sn_hook", None)
        if hook is not None:
            # spectral normalisation (round 4): torch's forward pre-hook was taken off the module by _conv(); the same arithmetic
            # runs here -- on the fused kernels (ops/spectral.py) in the case the trainer is in, through torch's own
            # SpectralNorm.compute_weight otherwise (eval mode, CPU, H3D_DISC_SN=torch)
            if os.environ.get("H3D_DISC_SN", "hip") != "torch" and spectral_ops.supported(self, hook):
                setattr(self, hook.name, spectral_ops.spectral_weight(self, hook))
            else:
                hook(self, None)
        if os.environ.get("H3D_DISC_CONV", "hip") != "torch":
            if torch.is_autocast_enabled() and x.is_cuda:
                # AMP tier (round 4): under float16 autocast the layer stays on the native kernels -- activations and their
                # gradients travel as f16 (h3d_conv_x3_f16 / h3d_conv_wgrad_x3_f16), the weights stay fp32; any other autocast
                # type goes to the library
                if torch.get_autocast_dtype("cuda") == torch.float16:
                    xh = x.half()
                    if conv_ops.supported(xh, self.weight):
                        return conv_ops.conv2d(xh, self.weight, self.bias)
            elif conv_ops.supported(x, self.weight):
                return conv_ops.conv2d(x, self.weight, self.bias)
            if x.is_cuda and not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") != torch.float16):
                # round 6: no silent library path on the GPU -- a call the native kernels do not cover (a kernel size other than
                # 1 / 3, a dtype other than fp32 / f16) is an error, as it is in the generator; H3D_DISC_CONV=torch opts into
                # torch's convolution for the whole discriminator.  (CPU tensors run torch's convolution: the world-2 gloo tests of
                # the trainers' control flow need the module on the host; autocast types other than float16 go to the library, above.)
                from ... import _lib
                raise _lib.H3DError(f"UNetDiscriminator: {tuple(x.shape)} {x.dtype} * {tuple(self.weight.shape)} is not covered by "
                                    "h3d_conv_x3 (kernel sizes 1 and 3, fp32 / f16 activations); set H3D_DISC_CONV=torch to run the "
                                    "discriminator on torch's convolutions")
        return super().forward(x)


def _conv(cin, cout, k, spectral):
    conv = Conv2d(cin, cout, k, 1, k // 2)
    if not spectral:
        return conv
    conv = nn.utils.spectral_norm(conv)           # parameters weight_orig, buffers weight_u / weight_v, state_dict hooks: torch's
    # ... but its forward pre-hook (13 small launches per layer and pass) is called from Conv2d.forward instead, which can then
    # run the fused kernels; the state_dict layout and the load / save hooks stay torch's
    for key, fn in list(conv._forward_pre_hooks.items()):
        if type(fn).__name__ == "SpectralNorm":
            del conv._forward_pre_hooks[key]
            conv._sn_hook = fn
    return conv


class ResBlock(nn.Module):
    """dx = conv2(lrelu(conv1([up](lrelu(x)))))  (+ average pooling when going down), shortcut = [resample +] [1x1 conv]."""

    def __init__(self, fin, fout, up_or_down, first=False, **kwargs):
        super().__init__()
        self.up_or_down, self.first = up_or_down, first
        sn = not kwargs.get("disable_spectral_norm", False)
        if first:
            self.conv1 = _conv(fin, fout, 3, sn)                       # no activation in front of the very first conv
        else:
            self.conv1 = _Slot(2 if up_or_down > 0 else 1, _conv(fin, fout, 3, sn))
        self.conv2 = _Slot(1, _conv(fout, fout, 3, sn))
        self.conv_s = _conv(fin, fout, 1, sn) if fin != fout else None

    def _resample(self, x):
        if self.up_or_down > 0:
            return F.interpolate(x, scale_factor=2, mode="nearest")
        if self.up_or_down < 0:
            return F.avg_pool2d(x, 2)
        return x

    def forward(self, x):
        """Same function as the reference's ResBlock.forward (unet_discriminators.py:48-72), evaluated in an order that does less
        work (round 4); every rewrite is exact algebra on the reference's graph:
          * up blocks: a 1x1 convolution commutes with nearest-neighbour upsampling (each output pixel is a function of one input
            pixel), so the shortcut is up(conv_s(x)) instead of conv_s(up(x)): a quarter of the convolution, fout instead of fin
            channels through the resampler -- bit for bit the same values;
          * down blocks (all but the first): average pooling is linear, so pool(conv_s(x)) + pool(conv2(..)) is one pooling of
            the sum (differs from the reference's two by fp32 rounding order, ~1e-7).
        The first block keeps the reference's order (pool BEFORE its 1x1 convolution: already the cheap one).
        The resampling / activation glue runs on two fused kernels (ops/pool_up.py: up(lrelu(x)) in one pass, up(s) + d,
        avgpool(s + d) -- each the other's adjoint, so the R1 double backward stays on them); `H3D_DISC_GLUE=torch` keeps
        F.leaky_relu / F.interpolate / F.avg_pool2d."""
        fused = x.is_cuda and os.environ.get("H3D_DISC_GLUE", "hip") != "torch" and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
        if self.first and self.up_or_down >= 0:
            fused = False       # the fused first-block glue IS the average pooling: a first block that does not go down keeps _resample
        d = x
        if not self.first:
            if self.up_or_down > 0 and fused and pool_up.supported(d):
                d = pool_up.lrelu_up2(d, 0.2)
            else:
                d = F.leaky_relu(d, 0.2)
                if self.up_or_down > 0:
                    d = self._resample(d)
        d = self.conv2(F.leaky_relu(self.conv1(d), 0.2))
        if self.first:
            s = pool_up.avgpool2(x) if fused and pool_up.supported(x) else self._resample(x)
            if self.conv_s is not None:
                s = self.conv_s(s)
            return s + (pool_up.avgpool2(d) if fused and pool_up.supported(d) else self._resample(d))
        s = x if self.conv_s is None else self.conv_s(x)
        if self.up_or_down > 0:
            return pool_up.up2_add(s, d) if fused and pool_up.supported(s, d) else self._resample(s) + d
        if self.up_or_down < 0:
            return pool_up.avgpool2_sum(s, d) if fused and pool_up.supported(s, d) else self._resample(s + d)
        return s + d


class UNetDiscriminator(nn.Module):

    def __init__(self, **kwargs):
        super().__init__()
        self.epoch = 0
        self.step = 0
        self.semantic_dim = kwargs.get("semantic_dim", 0)
        self.label_dim = kwargs.get("label_dim", 0)
        self.latent_dim = kwargs["latent_dim"]
        self.output_dim = self.semantic_dim + self.label_dim
        H, W = kwargs["gen_height"], kwargs["gen_width"]
        n = min(kwargs.get("discriminator_blocks", 6), int(math.log2(max(H, W))) - 1)
        self.num_blocks = n
        ch = [6 if kwargs.get("dual_discrimination", False) else 3, 128, 128, 256, 256, 512, 512, 512, 512]
        self.channels = ch
        self.body_down = nn.ModuleList(ResBlock(ch[i], ch[i + 1], -1, first=(i == 0), **kwargs) for i in range(n))
        ups = [ResBlock(ch[n], ch[n - 1], 1, **kwargs)]
        ups += [ResBlock(2 * ch[n - i], ch[n - i - 1], 1, **kwargs) for i in range(1, n - 1)]
        ups.append(ResBlock(2 * ch[1], 64, 1, **kwargs))
        self.body_up = nn.ModuleList(ups)
        self.layer_up_last = Conv2d(64, 1, 1)
        self.output_layer = Conv2d(64, self.output_dim, 1)
        self.latent_layer = nn.Conv2d(ch[n], self.latent_dim, (H // 2 ** n, W // 2 ** n))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                w = m.weight_orig if hasattr(m, "weight_orig") else m.weight
                nn.init.kaiming_normal_(w, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        with torch.no_grad():
            self.output_layer.weight *= 0.25

    def forward(self, images, conditions, alpha, **kwargs):
        """images [B, 3|6, H, W] -> {"prediction" [B,1,H,W], "latents" [B,latent_dim], "segments" [B,label_dim,H,W]
        (, "semantics" [B,semantic_dim,H,W])}.  `conditions` and `alpha` are accepted and unused, as in the reference."""
        x, skips = images, []
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)        # the native convolutions are channels-last throughout
        for blk in self.body_down:
            x = blk(x)
            skips.append(x)
        if min(x.shape[2:4]) > 1:
            latents = self.latent_layer(x).view(x.shape[0], self.latent_dim)
        else:
            latents = x.new_zeros((x.shape[0], self.latent_dim))
        x = self.body_up[0](x)
        for i in range(1, len(self.body_up)):
            x = self.body_up[i](torch.cat((skips[-i - 1], x), dim=1))
        prediction = self.layer_up_last(x)
        y = self.output_layer(x)
        out = {"prediction": prediction, "latents": latents, "segments": y[:, self.semantic_dim:]}
        if self.semantic_dim > 0:
            out["semantics"] = y[:, :self.semantic_dim]
        return out
