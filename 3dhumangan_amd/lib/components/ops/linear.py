"""y = x W^T + b on hand-written matrix-core kernels: the weight gradient on the split-K kernel of csrc/wgrad_x3.hip, and (round
3) forward and data gradient on h3d_conv_x3 as a 1x1 convolution over the rows.

Forward / data gradient ([M, C] x [C, C'] with M ~ 0.5 M rows): hipBLASLt runs them in fp32 at 131 TFLOP/s (83 % of the fp32 matrix
peak); h3d_conv_x3 evaluates the same contraction with split bf16 operands (three 16-bit products, ~2e-5) at ~230 TFLOP/s, bias
fused, when both widths are multiples of 64 (`H3D_LINEAR=library` keeps the library GEMM).  The WEIGHT gradient dW = dY^T X
contracts over the M rows and produces a tiny output -- the library runs it at ~50 TFLOP/s, h3d_wgrad_x3 streams both operands once
(HBM-bound) and returns the bias gradient from the same pass.  Used by lib/generators/differentiable.py for every layer with enough
rows; anything else (few rows, half-precision autocast inputs, odd widths, CPU tensors) is F.linear."""
import os
import weakref

import torch
import torch.nn.functional as F

from .... import _lib

MIN_ROWS = int(os.environ.get("H3D_WGRAD_MIN_ROWS", 16384))      # below this the library GEMM is as good
SMALL_ROWS = int(os.environ.get("H3D_AMP_FP32_ROWS", 64))            # AMP: layers with fewer rows than this run in fp32 (0: autocast decides)
ENABLED = os.environ.get("H3D_WGRAD", "x3") == "x3"
NATIVE_GEMM = os.environ.get("H3D_LINEAR", "x3") == "x3"
FUSED_ADD = os.environ.get("H3D_LINEAR_ADD", "fused") != "torch"          # residual addend in the GEMM epilogue (round 6; A/B switch)
AMP_NATIVE_GEMM = os.environ.get("H3D_AMP_LINEAR", "library") == "x3"      # AMP forward / data gradient: library f16 GEMM by default
PAD_ODD_WIDTH = os.environ.get("H3D_LINEAR_PAD", "1") != "0"                 # odd input widths padded for the weight-gradient kernel (round 6; A/B switch)
FUSED_MOMENTS = os.environ.get("H3D_FUSED_MOMENTS", "1") != "0"             # BatchNorm moments from the GEMM's accumulators (round 6; A/B switch)
# autocast dense layers WITH a residual addend on the own f16 GEMM (addend in the epilogue): opt-in -- measured slower in the iteration
# (106.0 -> 110.4 ms, profiles/r6_ab_amp_linear_add_not_kept.txt): in situ the own GEMM loses more to the library's than the sum's pass costs
AMP_ADD_NATIVE = os.environ.get("H3D_AMP_LINEAR_ADD", "library") == "x3"
# ... under float16 autocast too, on the own f16 GEMM: opt-in.  Same lease (profiles/r6_ab_amp_fused_moments_not_kept.txt): AMP iteration
# 109.7 -> 111.8 ms -- the own f16 GEMM takes 200 us where the library takes 140, and the moments pass it saves reads a tensor the
# infinity cache still holds
AMP_FUSED_MOMENTS = os.environ.get("H3D_AMP_FUSED_MOMENTS", "0") == "1"


_half_cache = {}


def _half_cached(t):
    """t.half() for a parameter, cast once per parameter VERSION (the optimiser's in-place update bumps it) instead of once per
    call: a dense layer's weight is used by the D step's generator forward, the G step's forward and its data gradient -- three
    casts and three tiny launches per layer and iteration otherwise (torch's autocast keeps the same kind of cache)."""
    if t is None or t.dtype == torch.float16:
        return t
    e = _half_cache.get(id(t))
    if e is not None and e[0] == t._version and e[1]() is t:
        return e[2]
    h = t.detach().half()
    if len(_half_cache) > 1024:                       # temporaries (a scaled weight built per call) leave dead entries behind
        for k in [k for k, v in _half_cache.items() if v[1]() is None]:
            del _half_cache[k]
    _half_cache[id(t)] = (t._version, weakref.ref(t), h)
    return h


def _as_image(t2):
    """[M, C] rows (unit column stride) -> the same memory as a [1, C, 1, M] channels-last "image" for h3d_conv_x3."""
    M, C = t2.shape
    ld = t2.stride(0)
    return torch.as_strided(t2, (1, C, 1, M), (M * ld, 1, M * ld, ld))


def gemm_x3(x2, w, bias=None, transposed=False, add=None, moments=False):
    """x2 [M, Ci] @ w[Co, Ci]^T (+ bias) (+ add [M, Co]) -> [M, Co]; transposed: x2 [M, Co] @ w[Co, Ci] -> [M, Ci].  Split-bf16
    matrix-core kernel; the addend (a residual connection) joins in the epilogue.  moments: -> (y, partial [rows, 2, Co]), the
    column sums of y and y^2 per workgroup, from the accumulators (h3d_conv_x3_moments)."""
    from . import conv
    # owner=w: the packed weight stream is cached on the caller's (long-lived) weight, not on this per-call view of it
    y = conv._run_conv(_as_image(x2), w.detach()[:, :, None, None], bias, transposed=transposed,
                       owner=w if w.dtype == torch.float32 else None, add=None if add is None else _as_image(add), moments=moments)
    if moments:
        return y[0].permute(0, 2, 3, 1).reshape(x2.shape[0], -1), y[1]
    return y.permute(0, 2, 3, 1).reshape(x2.shape[0], -1)


def _native_ok(Co, Ci):
    from . import conv
    return NATIVE_GEMM and Co % 64 == 0 and Ci % 64 == 0 and conv.tiling(Ci, Co) is not None and conv.tiling(Co, Ci) is not None


def wgrad_x3(dy, x, with_bias=False):
    """dy [M, Co], x [M, Ci] row-major (row stride >= width, multiple of 4), both fp32 or (AMP tier) both f16 -> dy^T x [Co, Ci]
    fp32; with_bias: also the column sums of dy [Co] (the bias gradient of the same layer, from the same pass over dy) -> (dw, db)."""
    _lib.need_cuda(dy, x)
    if dy.dtype != x.dtype:
        dy, x = dy.half(), x.half()
    M, Co = dy.shape
    Ci = x.shape[1]
    lib = _lib.load()
    slices = lib.h3d_wgrad_x3_slices(M, Co, Ci)
    partial = torch.empty((slices, Co, Ci), device=dy.device, dtype=torch.float32)
    colsum = torch.empty((slices, Co), device=dy.device, dtype=torch.float32) if with_bias else None
    entry = lib.h3d_wgrad_x3_bias if dy.dtype == torch.float32 else lib.h3d_wgrad_x3_bias_f16
    rc = entry(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(partial), _lib.ptr(colsum), M, Co, Ci, dy.stride(0), x.stride(0),
                               slices, _lib.stream_handle())
    _lib.check(rc, "h3d_wgrad_x3")
    if slices == 1:
        return (partial[0], colsum[0]) if with_bias else partial[0]
    from . import conv
    return conv.reduce_slices(partial, colsum, 1, slices, Co, Ci, (Co, Ci))      # both sums in one launch


def wgrad_narrow(wide, narrow, wide_sum=False, narrow_sum=False):
    """wide [M, C], narrow [M, n] (n <= 4), both fp32 or (AMP tier) both f16 -> narrow^T wide [n, C] fp32 (streams `wide` once;
    csrc/wgrad_narrow.hip).  wide_sum (n <= 3): also the column sums of `wide` [C]; narrow_sum: also those of `narrow` [n] -- the
    bias gradient of the layer, from the same pass.  -> dw, or (dw, sum)."""
    _lib.need_cuda(wide, narrow)
    if wide.dtype != narrow.dtype:
        wide, narrow = wide.half(), narrow.half()
    M, C = wide.shape
    n = narrow.shape[1]
    narrow = narrow.contiguous()
    lib = _lib.load()
    ones = int(bool(wide_sum))
    nblk = (M + lib.h3d_wgrad_narrow_rows() - 1) // lib.h3d_wgrad_narrow_rows()
    partial = torch.empty((nblk, n + ones, C), device=wide.device, dtype=torch.float32)
    colsum = torch.empty((nblk, 4), device=wide.device, dtype=torch.float32) if narrow_sum else None
    rc = lib.h3d_wgrad_narrow_sums(_lib.ptr(wide), _lib.ptr(narrow), _lib.ptr(partial), _lib.ptr(colsum), M, C, wide.stride(0), n,
                                   ones, int(wide.dtype == torch.float16), _lib.stream_handle())
    _lib.check(rc, "h3d_wgrad_narrow")
    out = partial.sum(dim=0)
    if wide_sum:
        return out[:n], out[n]
    if narrow_sum:
        return out, colsum.sum(dim=0)[:n]
    return out


def _rows(t):
    """[..., C] -> a [M, C] view with unit column stride and a row stride that is a multiple of 4, or a contiguous copy."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or t2.stride(0) % (4 if t2.dtype == torch.float32 else 8) or t2.stride(0) < t2.shape[1]:
        t2 = t2.contiguous()
    return _lib.aligned16(t2)          # a contiguous view at an odd storage offset is copied (.contiguous() would return it as is)


class _LinearX3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, add=None, moments=False):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        if _native_ok(*w.shape):
            a2 = None if add is None else _rows(add.detach())
            if moments:           # (y, partial moments of y): the second output carries no gradient
                y, partial = gemm_x3(_rows(x), w, b, add=a2, moments=True)
                ctx.mark_non_differentiable(partial)
                return y.view(*x.shape[:-1], w.shape[0]), partial
            return gemm_x3(_rows(x), w, b, add=a2).view(*x.shape[:-1], w.shape[0])
        if moments:
            raise ValueError("moments need the native GEMM")
        y = F.linear(x, w, b)
        return y if add is None else y + add

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        dy2 = _rows(dy)
        if ctx.needs_input_grad[0]:
            dx = gemm_x3(dy2, w, transposed=True).view(*dy.shape[:-1], w.shape[1]) if _native_ok(*w.shape) else dy @ w
        if ctx.needs_input_grad[1]:
            Co, Ci = w.shape
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if Co <= 4:
                dw = wgrad_narrow(_rows(x), dy2, narrow_sum=want_db)   # [Co, Ci] (, column sums of dy: the bias gradient)
                if want_db:
                    dw, db = dw
            elif Ci <= 3 and want_db:
                dw, db = wgrad_narrow(dy2, _rows(x), wide_sum=True)  # [Ci, Co], column sums of dy
                dw = dw.t()
            elif Ci <= 4:
                dw = wgrad_narrow(dy2, _rows(x)).t()                 # [Ci, Co] -> [Co, Ci]
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                dw, db = wgrad_x3(dy2, _rows(x), with_bias=True)      # the bias gradient rides along: no second pass over dy
            else:
                dw = wgrad_x3(dy2, _rows(x))
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(dim=0)
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None), None    # the addend's gradient is the output's


class _LinearAmp(torch.autograd.Function):
    """The dense layer under float16 autocast (AMP tier, round 4; reference: nn.Linear / 1x1 convs inside torch.cuda.amp.autocast):
    forward and data gradient are the library's f16 GEMMs on f16 activations (HBM-bound there: 126 us for 0.5 M x 256 x 256), or
    h3d_conv_x3_f16 with H3D_AMP_LINEAR=x3; the weight gradient (tall-skinny TN, the shape the library is slow at: 4 ms for a
    3 x 256 result) is h3d_wgrad_x3 / h3d_wgrad_narrow on the f16 operands as they are -- fp32 result, no casts; the weight
    stays fp32."""

    @staticmethod
    def forward(ctx, x, w, b, add=None, moments=False):
        xh = x.half()
        ctx.save_for_backward(xh, w)
        ctx.has_bias = b is not None
        if moments:
            # a layer in front of a SPADE (round 6): the own f16 GEMM (one weight plane = autocast's arithmetic) is slower than the
            # library's (200 vs 140 us at 0.5 M x 256 x 256) but hands over the batch moments from its accumulators (a 94 us pass
            # otherwise) and takes the residual addend in fp32 before the one rounding (another pass and another rounding otherwise)
            a2 = None if add is None else _rows(add.detach().half())
            y, partial = gemm_x3(_rows(xh), w, b, add=a2, moments=True)
            ctx.mark_non_differentiable(partial)
            return y.view(*x.shape[:-1], w.shape[0]), partial
        if (AMP_NATIVE_GEMM or (add is not None and AMP_ADD_NATIVE)) and _native_ok(*w.shape):
            # opt-in paths: the own f16 GEMM, the addend (if any) in its epilogue -- added in fp32, one rounding
            y = gemm_x3(_rows(xh), w, b, add=None if add is None else _rows(add.detach().half()))
            return y.view(*x.shape[:-1], w.shape[0])
        y = F.linear(xh, _half_cached(w), _half_cached(b))
        return y if add is None else y + add

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        xh, w = ctx.saved_tensors
        dyh = dy.half()
        dx = dw = db = None
        dy2, x2 = _rows(dyh), _rows(xh)
        if ctx.needs_input_grad[0]:
            if AMP_NATIVE_GEMM and _native_ok(*w.shape):
                dx = gemm_x3(dy2, w, transposed=True).view(*dy.shape[:-1], w.shape[1])
            else:
                dx = dyh @ _half_cached(w)
        if ctx.needs_input_grad[1]:
            Co, Ci = w.shape
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if Co <= 4:
                dw = wgrad_narrow(x2, dy2, narrow_sum=want_db)       # [Co, Ci] (, column sums of dy: the bias gradient)
                if want_db:
                    dw, db = dw
            elif Ci <= 3 and want_db:
                dw, db = wgrad_narrow(dy2, x2, wide_sum=True)        # [Ci, Co], column sums of dy
                dw = dw.t()
            elif Ci <= 4:
                dw = wgrad_narrow(dy2, x2).t()                       # [Ci, Co] -> [Co, Ci]
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                dw, db = wgrad_x3(dy2, x2, with_bias=True)
            else:
                dw = wgrad_x3(dy2, x2)
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(dim=0, dtype=torch.float32)
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None), None


def linear(x, w, b=None, add=None, moments=False):
    """F.linear(x, w, b) (+ add: a residual connection, fused into the native GEMM's epilogue where that runs); the weight gradient
    goes to the HIP kernel when the problem is one it is built for.
    moments=True: -> (y, partial): partial [rows, 2, Co] fp32 holds per-workgroup column sums of y and y^2 taken from the GEMM's
    accumulators (h3d_conv_x3_moments; spade_norm_act(.., moments=partial) then skips its own pass over y), or None where the call
    does not run on the native fp32 GEMM (the SPADE computes them itself)."""
    if moments:
        Co, Ci = w.shape
        rows = x.numel() // max(Ci, 1)
        if (AMP_FUSED_MOMENTS and FUSED_MOMENTS and ENABLED and x.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.float16 and w.dtype == torch.float32 and rows >= MIN_ROWS and _native_ok(Co, Ci) and Co >= 32 and Ci >= 32
                and (add is None or add.shape == x.shape[:-1] + (Co,))):
            return _LinearAmp.apply(x, w, b, add, True)
        ok = (FUSED_MOMENTS and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled()
              and rows >= MIN_ROWS and _native_ok(Co, Ci) and (add is None or (FUSED_ADD and add.dtype == torch.float32
                                                                              and add.shape == x.shape[:-1] + (Co,))))
        if ok and ENABLED and torch.is_grad_enabled() and w.requires_grad and Co >= 32 and Ci >= 32:
            return _LinearX3.apply(x, w, b, add, True)
        if ok and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (add is not None and add.requires_grad))):
            y, partial = gemm_x3(_rows(x), w, b, add=None if add is None else _rows(add), moments=True)
            return y.view(*x.shape[:-1], Co), partial
        return linear(x, w, b, add), None
    Co, Ci = w.shape
    rows = x.numel() // max(Ci, 1)
    if (PAD_ODD_WIDTH and ENABLED and add is None and x.is_cuda and w.dtype == torch.float32 and torch.is_grad_enabled() and w.requires_grad
            and not x.requires_grad and rows >= MIN_ROWS and Co % 8 == 0 and Co >= 32 and Ci > 4 and Ci % 8
            and (x.dtype == torch.float32 or torch.is_autocast_enabled())):
        # an input width the weight-gradient kernel does not take (the field's 31 geometry features, map3d_layers / smpl.py:210-249):
        # one zero column more and the layer's weight gradient leaves the library's tall-skinny TN GEMM (0.99 ms at 0.59 M rows x 31 ->
        # 256) for h3d_wgrad_x3 (round 6).  The input needs no gradient (it is data), the weight's comes back through the pad.
        pad = 8 - Ci % 8
        return linear(F.pad(x, (0, pad)), F.pad(w, (0, pad)), b)
    if (add is not None and AMP_ADD_NATIVE and FUSED_ADD and ENABLED and x.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.float16 and w.dtype == torch.float32 and rows >= MIN_ROWS and _native_ok(Co, Ci)
            and Co >= 32 and Ci >= 32 and add.shape == x.shape[:-1] + (Co,)):
        return _LinearAmp.apply(x, w, b, add, False)
    if add is not None:
        fp32 = (FUSED_ADD and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and add.dtype == torch.float32
                and not torch.is_autocast_enabled() and rows >= MIN_ROWS and _native_ok(Co, Ci) and add.shape == x.shape[:-1] + (Co,))
        if fp32 and ENABLED and torch.is_grad_enabled() and w.requires_grad and Co >= 32 and Ci >= 32:
            return _LinearX3.apply(x, w, b, add)
        if fp32 and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or add.requires_grad)):
            return gemm_x3(_rows(x), w, b, add=_rows(add)).view(*x.shape[:-1], Co)
        return linear(x, w, b) + add
    if (ENABLED and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16
            and torch.is_grad_enabled() and w.requires_grad and w.dtype == torch.float32 and rows >= MIN_ROWS
            and ((Co % 8 == 0 and Ci % 8 == 0 and Co >= 32 and Ci >= 32) or (Co <= 4 and Ci >= 32) or (Ci <= 4 and Co >= 32))):
        return _LinearAmp.apply(x, w, b, None, False)
    if (ENABLED and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and torch.is_grad_enabled()
            and w.requires_grad and not torch.is_autocast_enabled() and rows >= MIN_ROWS
            and ((Co % 4 == 0 and Ci % 4 == 0 and Co >= 32 and Ci >= 32)           # h3d_wgrad_x3
                 or (Co <= 4 and Ci >= 32) or (Ci <= 4 and Co >= 32))):             # h3d_wgrad_narrow (heads, ToRGB, coordinates)
        return _LinearX3.apply(x, w, b, None)
    if (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled() and rows >= MIN_ROWS
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)) and _native_ok(Co, Ci)):
        return gemm_x3(_rows(x), w, b).view(*x.shape[:-1], Co)          # nothing to record (the D step's generator forward)
    if x.is_cuda and rows < SMALL_ROWS and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16:
        # a handful of rows (per-sample style vectors: the constant SPADEs' modulation, the shared layers' offsets) under float16
        # autocast: computed in fp32 (round 6).  Autocast would convert both parameters on every call -- the weight handed in is a
        # view of the parameter, which its cache never holds -- and their gradients back: 4 launches per layer and pass, ~350 per
        # config-4 iteration, for products of 4 x 128 x 256.
        with torch.autocast("cuda", enabled=False):
            return F.linear(x.float(), w.float(), None if b is None else b.float())
    return F.linear(x, w, b)
