"""conv2d (3x3 / 1x1, stride 1, "same" padding) on the hand-written matrix-core kernels of csrc/conv_x3.hip / wgrad_x3.hip,
differentiable to any order -- what the discriminator (lib/discriminators/unet_discriminators.py; reference ResBlock,
unet_discriminators.py:8-72) runs its convolutions on, including the double backward of the R1 penalty
(lib/trainers/phase_trainer.py:259-294).

Three primitives, closed under differentiation (each backward is written with the other two, so autograd can differentiate
the backward pass again -- no graph through a library convolution):
    C (x, W)  = conv(x, W)                    dC/dx  = Ct(g, W)     dC/dW  = Cw(x, g)
    Ct(g, W)  = conv_transpose(g, W)          dCt/dg = C (h, W)     dCt/dW = Cw(h, g)
    Cw(x, g)  = weight gradient [Co,Ci,k,k]   dCw/dx = Ct(g, V)     dCw/dg = C (x, V)
C and Ct are one kernel (h3d_conv_x3): Ct runs it on the flipped, transposed weights; Cw is h3d_conv_wgrad_x3.  Activations are
channels-last fp32 (torch.channels_last memory format of the logical NCHW tensors); weights are packed per call on the device
(a permute and two casts) -- they change every optimiser step and carry the spectral normalisation's graph.
"""
import ctypes
import os
import weakref

import torch

from .... import _lib


def tiling(cin, cout):
    """(NT, output blocks, k-steps per chunk, chunks) of h3d_conv_x3 for cin -> cout, or None when unsupported."""
    out = (ctypes.c_int * 4)()
    if _lib.load().h3d_conv_x3_tiling(int(cin), int(cout), out):
        return None
    return tuple(out)


def _native(ci, co):
    return ci % 64 == 0 and co % 64 == 0 and tiling(ci, co) is not None


def _up64(n):
    return (n + 63) // 64 * 64


def supported(x, weight):
    """The native path takes fp32 CUDA tensors, k in {1, 3} (square).  Channel counts that are multiples of 64 run as they are;
    any other count (the RGB stem, the 1- and label_dim-channel heads, odd widths of the modulated convolutions) is zero-padded
    to the next multiple of 64 by conv2d."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.float16) and weight.dtype == torch.float32 and weight.dim() == 4):
        return False                      # f16 activations: the AMP tier (h3d_conv_x3_f16); weights stay fp32
    co, ci, kh, kw = weight.shape
    if not (kh == kw and kh in (1, 3) and x.shape[1] == ci):
        return False
    return _native(_up64(ci), _up64(co))


_stream_cache = {}
# AMP tier: f16 weight planes of the convolution kernels.  1 (default, round 5) = the weight rounded to f16 once, as the reference's
# autocast rounds it: one matrix product per weight, half the weight stream (h3d_conv_x3_f16x1); 2 = f16 hi + lo (the weight to
# max(2^-22 |W|, 2^-25): more precise than what it is compared with, twice the matrix work) -- the opt-in tier.
AMP_WEIGHT_PLANES = int(os.environ.get("H3D_AMP_WEIGHT_PLANES", "1"))
FUSED_PAD = os.environ.get("H3D_CONV_PAD", "fused") != "torch"             # channel padding on h3d_pad_channels_cl (round 6)
FUSED_REDUCE = os.environ.get("H3D_WGRAD_REDUCE", "fused") != "torch"      # the slices' sum on h3d_wgrad_reduce (round 6)


def pack_stream(w, transposed=False, half=False, owner=None, planes=2, nt=0):
    """w [Co, Ci, k, k] fp32 (device) -> the bf16 hi/lo weight stream of h3d_conv_x3 (int16 bit patterns; include/h3d.h), one
    kernel launch (h3d_conv_x3_pack).  transposed: the stream of w's backward-data convolution (Ci -> Co channels swapped,
    taps flipped) instead.  Cached per tensor OBJECT and version: the same (spectrally normalised) weight is convolved with in the
    forward, again in the R1 double backward, and its transposed stream in both backward passes -- one pack each instead of one
    per call.  `owner`: the long-lived tensor `w` is a per-call view / reshape of (the dense layers hand in
    ``weight.detach()[:, :, None, None]``, a new object every call): the cache is keyed on the owner's identity and version, so
    those calls hit too.  Inference tensors (torch.inference_mode) have no version counter: they are packed every call.
    half: the f16 hi/lo stream of h3d_conv_x3_f16.  nt: tiles per output block the stream is laid out for (h3d_conv_x3_nt_for;
    0 = the default blocking) -- part of the cache key."""
    ref = w if owner is None else owner
    cacheable = not ref.is_inference()
    one = bool(half) and planes == 1
    key = (id(ref), bool(transposed), bool(half), tuple(w.shape), one, int(nt))
    if cacheable:
        e = _stream_cache.get(key)
        if e is not None and e[0] == ref._version and e[1]() is ref:
            return e[2]
    wd = w.detach().contiguous()
    co, ci, k, _ = wd.shape
    out = torch.empty((1 if one else 2) * wd.numel(), device=wd.device, dtype=torch.int16)
    rc = _lib.load().h3d_conv_x3_pack_nt(_lib.ptr(wd), _lib.ptr(out), ci if transposed else co, co if transposed else ci, k, int(transposed),
                                         2 if one else 1 if half else 0, int(nt), _lib.stream_handle())
    _lib.check(rc, "h3d_conv_x3_pack_nt")
    if not cacheable:
        return out
    if len(_stream_cache) > 32:            # dead entries (their tensors are gone) hold device memory: drop them early
        for kk in [kk for kk, v in _stream_cache.items() if v[1]() is None]:
            del _stream_cache[kk]
    _stream_cache[key] = (ref._version, weakref.ref(ref), out)
    return out


def pack_stream_torch(w):
    """The same stream with tensor operations (reference for tests/test_gpu_conv.py)."""
    co, ci, k, _ = w.shape
    NT, nblk, KSC, nch = tiling(ci, co)
    t = w.detach().reshape(nblk, NT, 32, nch, KSC, 2, 8, k * k)          # ob, nt, j, chunk, ks, h, e, tap
    t = t.permute(0, 7, 3, 4, 1, 5, 2, 6)                                 # ob, tap, chunk, ks, nt, h, j, e
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], dim=5).contiguous().view(torch.int16)    # ob, tap, chunk, ks, nt, (hi|lo), h, j, e


_copied = {}          # H3D_CONV_DEBUG=1: (shape, strides) of tensors that had to be copied to channels-last, with counts


def _rows(x):
    """x [B, C, H, W] -> (tensor whose memory is pixel-major rows of C floats, row stride in floats).  Channels-last tensors and
    channel slices of channels-last tensors (what the backward of a skip concatenation hands out) are taken as they are;
    anything else is copied to channels-last."""
    x = x.detach()
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    ld = sw
    gran = 4 if x.dtype == torch.float32 else 8           # 16-byte accesses: 4 floats or 8 halves
    if (sc == 1 and ld >= C and ld % gran == 0 and sh == W * ld and (sb == H * W * ld or B == 1) and x.data_ptr() % 16 == 0):
        return x, ld
    if os.environ.get("H3D_CONV_DEBUG"):
        key = (tuple(x.shape), tuple(x.stride()))
        _copied[key] = _copied.get(key, 0) + 1
    return _lib.aligned16(x.contiguous(memory_format=torch.channels_last)), C


def _run_conv(x, w, bias=None, transposed=False, owner=None, add=None, moments=False):
    """x [B, Ci, H, W] (any layout), w [Co, Ci, k, k] -> [B, Co, H, W] channels-last; no autograd.  transposed: w is
    [Ci, Co, k, k] and the backward-data convolution of w runs instead (x has w's OUTPUT channel count).  add [B, Co, H, W] of x's
    type: added to the output in the kernel's epilogue (h3d_conv_x3_add: a residual connection without a pass of its own).
    moments: -> (out, partial [rows, 2, Co] fp32), the per-workgroup column sums of the stored output and of its square
    (h3d_conv_x3_moments: the BatchNorm statistic of the next SPADE from the accumulators)."""
    x, ldx = _rows(x)
    B, ci, H, W = x.shape
    k = w.shape[2]
    co = w.shape[1] if transposed else w.shape[0]
    wf = w.float()                         # fp32 weights come back as the same object: the cache key survives
    half = x.dtype == torch.float16
    lib = _lib.load()
    # blocking for THIS pixel count (round 6): the widest whose grid still fills the chip -- part of the stream's layout, so of its cache key
    nt = lib.h3d_conv_x3_nt_for(ci, co, B * H * W)
    stream = pack_stream(wf, transposed, half=half, owner=owner if (owner is not None and wf is w) else None, planes=AMP_WEIGHT_PLANES, nt=nt)
    out = torch.empty((B, co, H, W), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    b = None if bias is None else _lib.aligned16(bias.detach().float().contiguous())      # a slice of a larger bias vector may start anywhere
    mode = 0 if not half else (2 if AMP_WEIGHT_PLANES == 1 else 1)
    lda = 0
    if add is not None:
        if add.dtype != x.dtype or tuple(add.shape) != (B, co, H, W):
            raise ValueError(f"addend {tuple(add.shape)} {add.dtype} does not match the output {(B, co, H, W)} {x.dtype}")
        add, lda = _rows(add)
    partial = None
    if moments:
        per = lib.h3d_conv_x3_moment_rows()
        partial = torch.empty(((B * H * W + per - 1) // per, 2, co), device=x.device, dtype=torch.float32)
    # K-slices (round 6): a layer whose grid leaves CUs idle even at the narrowest blocking runs its tap x chunk loop in pieces
    slices = 1 if moments else lib.h3d_conv_x3_slices(ci, co, k, B * H * W, nt)
    work = torch.empty((slices, B * H * W, co), device=x.device, dtype=torch.float32) if slices > 1 else None
    rc = lib.h3d_conv_x3_ex(mode, _lib.ptr(x), _lib.ptr(stream), _lib.ptr(b), _lib.ptr(add), _lib.ptr(out), _lib.ptr(partial),
                            _lib.ptr(work), slices, B, H, W, ci, co, k, ldx, co, lda, nt, _lib.stream_handle())
    _lib.check(rc, "h3d_conv_x3_ex")
    return (out, partial) if moments else out


def _run_wgrad(x, g, k, with_bias=False):
    """x [B, Ci, H, W], g [B, Co, H, W] -> [Co, Ci, k, k] fp32; no autograd.  f16 operands (AMP) go to the _f16 entry point: both
    then travel as f16 (a mixed pair is brought to f16 first).  with_bias: also the bias gradient g.sum((0, 2, 3)) [Co] fp32, from
    the same pass over g (h3d_conv_wgrad_x3_bias) -> (dw, db)."""
    if x.dtype != g.dtype:
        x, g = x.half(), g.half()
    (x, ldx), (g, ldg) = _rows(x), _rows(g)
    B, ci, H, W = x.shape
    co = g.shape[1]
    lib = _lib.load()
    slices = max(1, lib.h3d_conv_wgrad_x3_slices(B, H, W, co, ci, k))
    partial = torch.empty((k * k, slices, co, ci), device=x.device, dtype=torch.float32)
    if with_bias:
        colsum = torch.empty((slices, co), device=x.device, dtype=torch.float32)
        rc = lib.h3d_conv_wgrad_x3_bias(_lib.ptr(g), _lib.ptr(x), _lib.ptr(partial), _lib.ptr(colsum), B, H, W, co, ci, k, ldg, ldx,
                                        slices, int(x.dtype == torch.float16), _lib.stream_handle())
    else:
        entry = lib.h3d_conv_wgrad_x3 if x.dtype == torch.float32 else lib.h3d_conv_wgrad_x3_f16
        rc = entry(_lib.ptr(g), _lib.ptr(x), _lib.ptr(partial), B, H, W, co, ci, k, ldg, ldx, slices, _lib.stream_handle())
    _lib.check(rc, "h3d_conv_wgrad_x3")
    return reduce_slices(partial, colsum if with_bias else None, k * k, slices, co, ci, (co, ci, k, k))


def reduce_slices(partial, colsum, taps, slices, co, ci, shape):
    """partial [taps, slices, co, ci] (, colsum [slices, co]) -> dw `shape` = [co, ci (, k, k)] (, db [co]): the slices' sum, the
    layout change and the bias gradient's sum in ONE launch (h3d_wgrad_reduce; torch: a reduction, a permuted copy and a second
    reduction -- three launches per weight gradient)."""
    if not FUSED_REDUCE:                  # H3D_WGRAD_REDUCE=torch: the round-5 tensor operations (A/B switch)
        dw = partial.view(taps, slices, co, ci).sum(dim=1).permute(1, 2, 0).contiguous().view(shape)
        return dw if colsum is None else (dw, colsum.sum(dim=0))
    dw = torch.empty(shape, device=partial.device, dtype=torch.float32)
    db = None if colsum is None else torch.empty((co,), device=partial.device, dtype=torch.float32)
    rc = _lib.load().h3d_wgrad_reduce(_lib.ptr(partial), _lib.ptr(colsum), _lib.ptr(dw), _lib.ptr(db), taps, slices, co, ci,
                                      _lib.stream_handle())
    _lib.check(rc, "h3d_wgrad_reduce")
    return dw if colsum is None else (dw, db)


# torch.autograd.grad(.., inputs=[x]) prunes NODES that do not lead to x, but inside a custom Function ctx.needs_input_grad is what the
# forward call saw: _Conv.backward would compute its weight and bias gradients in the R1 penalty's first-order pass
# (losses.r1_gradient: d prediction / d image, phase_trainer.py:259-283), where the engine drops them unread -- one full set of the
# discriminator's weight gradients per iteration.  A library convolution gets this pruning from the engine's output mask; here the
# caller says so.  Module-global on purpose: the backward pass runs on the autograd engine's device thread, inside the blocking call.
_INPUT_GRADS_ONLY = False


class input_grads_only:
    """with input_grads_only(): torch.autograd.grad(y, [x], ..) -- the native convolutions return no weight / bias gradients."""

    def __enter__(self):
        global _INPUT_GRADS_ONLY
        self.keep, _INPUT_GRADS_ONLY = _INPUT_GRADS_ONLY, True

    def __exit__(self, *exc):
        global _INPUT_GRADS_ONLY
        _INPUT_GRADS_ONLY = self.keep
        return False


def _transposed(w):
    return w.flip(2, 3).transpose(0, 1).contiguous()


class _Conv(torch.autograd.Function):
    """conv(x, w) (+ b, added by the kernel's epilogue: no separate pass over the output)."""

    @staticmethod
    def forward(ctx, x, w, b=None):
        ctx.save_for_backward(x, w)
        return _run_conv(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.to(x.dtype)                   # the cotangent of an f16 output is f16; the weight / bias gradients come out fp32
        gx = _ConvT.apply(g, w) if ctx.needs_input_grad[0] else None
        want_b = len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]
        gw = gb = None
        if _INPUT_GRADS_ONLY:
            return gx, None, None
        if ctx.needs_input_grad[1] and want_b:
            gw, gb = _ConvWB.apply(x, g, w.shape[2])          # the bias gradient rides along the weight-gradient pass over g
        elif ctx.needs_input_grad[1]:
            gw = _ConvW.apply(x, g, w.shape[2])
        elif want_b:
            gb = g.sum(dim=(0, 2, 3), dtype=torch.float32)
        return gx, gw, gb


class _ConvT(torch.autograd.Function):
    """conv_transpose(g, w): g [B, Co, H, W], w [Co, Ci, k, k] -> [B, Ci, H, W] (the data gradient of _Conv)."""

    @staticmethod
    def forward(ctx, g, w):
        ctx.save_for_backward(g, w)
        return _run_conv(g, w, transposed=True)

    @staticmethod
    def backward(ctx, h):
        g, w = ctx.saved_tensors
        h = h.to(g.dtype)
        gg = _Conv.apply(h, w) if ctx.needs_input_grad[0] else None
        gw = _ConvW.apply(h, g, w.shape[2]) if (ctx.needs_input_grad[1] and not _INPUT_GRADS_ONLY) else None
        return gg, gw


class _ConvW(torch.autograd.Function):
    """weight gradient of _Conv: x [B, Ci, H, W], g [B, Co, H, W] -> [Co, Ci, k, k]."""

    @staticmethod
    def forward(ctx, x, g, k):
        ctx.save_for_backward(x, g)
        return _run_wgrad(x, g, k)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        v = v.float()
        gx = _ConvT.apply(g, v) if ctx.needs_input_grad[0] else None
        gg = _Conv.apply(x, v) if ctx.needs_input_grad[1] else None
        return gx, gg, None


class _ConvWB(torch.autograd.Function):
    """_ConvW together with the bias gradient g.sum((0, 2, 3)): x, g -> ([Co, Ci, k, k], [Co]), one pass over g."""

    @staticmethod
    def forward(ctx, x, g, k):
        ctx.save_for_backward(x, g)
        return _run_wgrad(x, g, k, with_bias=True)

    @staticmethod
    def backward(ctx, v, vb):
        x, g = ctx.saved_tensors
        v = v.float()
        gx = _ConvT.apply(g, v) if ctx.needs_input_grad[0] else None
        gg = _Conv.apply(x, v) if ctx.needs_input_grad[1] else None
        if gg is not None and vb is not None:
            gg = gg + vb.to(gg.dtype).view(1, -1, 1, 1)          # d(sum over pixels of g) / dg
        return gx, gg, None


def _pad_channels(x, cop):
    """x [B, C, H, W] (any layout, fp32 / f16) -> [B, cop, H, W] channels-last with channels C .. cop zero; no autograd (h3d_pad_channels_cl)."""
    _lib.need_cuda(x)
    x = x.detach()
    B, C, H, W = x.shape
    if not FUSED_PAD:                      # H3D_CONV_PAD=torch: the round-5 tensor operations (A/B switch)
        x = x.contiguous(memory_format=torch.channels_last)
        out = torch.cat([x, x.new_zeros((B, cop - C, H, W)).contiguous(memory_format=torch.channels_last)], dim=1)
        return out.contiguous(memory_format=torch.channels_last)
    sb, sc, sh, sw = x.stride()
    if H > 1 and W > 1 and sh != W * sw:   # rows that do not follow each other: not a pixel-strided layout
        x = x.contiguous(memory_format=torch.channels_last)
        sb, sc, sh, sw = x.stride()
    sp = sh if W == 1 else sw              # a dimension of size one carries an arbitrary stride
    out = torch.empty((B, cop, H, W), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    rc = _lib.load().h3d_pad_channels_cl(_lib.ptr(x), _lib.ptr(out), B, C, cop, H * W, sb, sc, sp, int(x.dtype == torch.float16),
                                         _lib.stream_handle())
    _lib.check(rc, "h3d_pad_channels_cl")
    return out


class _PadChannels(torch.autograd.Function):
    """torch.cat([x, zeros], dim=1) in channels-last, one pass (h3d_pad_channels_cl); the adjoint of _NarrowChannels, and its
    backward is _NarrowChannels: the pair is closed under differentiation (the R1 double backward goes through the stem)."""

    @staticmethod
    def forward(ctx, x, cop):
        ctx.ci = x.shape[1]
        return _pad_channels(x, cop)

    @staticmethod
    def backward(ctx, g):
        return _NarrowChannels.apply(g, ctx.ci), None


class _NarrowChannels(torch.autograd.Function):
    """y[:, :co] whose gradient stays channels-last (torch's slice backward builds an NCHW-contiguous zero-padded tensor, which the
    next convolution would have to copy); differentiable again (_PadChannels)."""

    @staticmethod
    def forward(ctx, y, co):
        ctx.full = y.shape[1]
        return y[:, :co]

    @staticmethod
    def backward(ctx, g):
        return _PadChannels.apply(g, ctx.full), None


def conv2d(x, weight, bias=None):
    """F.conv2d(x, weight, bias, stride=1, padding=k // 2) on the native kernels (see `supported`); the bias is a broadcast add.
    A side with fewer than 64 channels is zero-padded to 64 (differentiably: a concatenation / a slice), which costs little
    where it happens -- 3 -> 128 at full resolution becomes the work of a 64 -> 128 layer, the heads that of 64 -> 64 1x1."""
    _lib.need_cuda(x, weight, bias)
    co, ci = weight.shape[:2]
    cip, cop = _up64(ci), _up64(co)
    if cip != ci:
        x = _PadChannels.apply(x, cip) if (x.requires_grad and torch.is_grad_enabled()) else _pad_channels(x, cip)
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, cip - ci))
    weight = weight.float()                 # parameters stay fp32 under autocast: the kernels split them to bf16 hi / lo themselves
    if cop != co:
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, cop - co))
    if (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)) and torch.is_grad_enabled():
        if cop == co:
            return _Conv.apply(x, weight, bias)
        # padded output channels: the bias travels zero-padded through the kernel's epilogue (and its gradient comes back from
        # the weight-gradient pass) instead of a broadcast add and a reduction over the sliced output
        y = _Conv.apply(x, weight, None if bias is None else torch.nn.functional.pad(bias.float(), (0, cop - co)))
    elif cop == co:
        return _run_conv(x, weight, bias)
    else:
        y = _run_conv(x, weight, None if bias is None else torch.nn.functional.pad(bias.float(), (0, cop - co)))
    return _NarrowChannels.apply(y, co) if y.requires_grad else y[:, :co]
