"""Float64 emulation of the "x2" split arithmetic of the matrix-core engines (csrc/field_x2.hip), for CPU tests.

A product W.x is evaluated as
    hi(W).hi(x)                                      one f16 MFMA (fp32 accumulate), exact products of f16 values
  + [ q6(hi(W)) . q6(lo(x)) + q6(lo(W)) . q6(hi(x)) ]  one block-scaled fp6 (e2m3) MFMA: the two cross terms, which only need a few
                                                     significant bits because they are 2^-11 of the main term
(lo.lo, 2^-22 relative, is dropped as in the x3 engines).  hi = f16 rounding, lo = the fp32 residual; q6 = round-to-nearest-even,
saturating e2m3 (1 sign, 2 exponent, 3 mantissa bits: 0, 0.125 .. 0.875, 1 .. 7.5) of the value times a power-of-two scale:
weights: one power-of-two scale alpha per (output row, 16-feature slot group): the largest with |hi| alpha <= 7.5 if no lo code
saturates (|lo| 2^12 alpha <= 7.5), else half of it; activations: static scales 4 (hi, |x| <= 1 after a sine: field engine) and
4 * 2^12 (lo), or, for unbounded activations (synthesis engine), a per-(sample, group) power of two from the largest |x| of
the group: the largest with |hi| / cs < 7.5.
"""
import numpy as np
import torch

_CODES = np.array([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                   2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5])


def q_e2m3(v):
    """Round-to-nearest-even, saturating fp6 e2m3 quantisation of a float64 tensor (values, not codes)."""
    a = v.abs().clamp(max=7.5)
    step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    q = torch.round(a / step) * step          # torch.round is half-to-even; code spacing doubles exactly at 2 and 4
    return torch.sign(v) * q.clamp(max=7.5)


def f16(v):
    return v.to(torch.float16).to(torch.float64)


def acc_k(ks, hh, e):
    """feature index of k-slot (lane half hh, element e) of k-step ks in accumulator-register order (csrc/field_x3.hip)."""
    return 32 * (ks // 2) + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hh


def slot_groups(K):
    """Index arrays of the 16-feature groups that share one fp6 block scale: (K-tile T, lane half h) -> features."""
    groups = []
    for T in range((K + 31) // 32):
        for hh in range(2):
            idx = [acc_k(2 * T + j, hh, e) for j in range(2) for e in range(8)]
            groups.append(torch.tensor([i for i in idx if i < K], dtype=torch.long))
    return [g for g in groups if len(g)]


def weight_alpha(Wh_g, Wl_g, rho=4096.0):
    """Block scale of one slot group: Wh_g, Wl_g [N, 16] -> alpha [N] (see the module docstring)."""
    wmax = Wh_g.abs().amax(dim=1)
    a2 = torch.where(wmax > 0, 2.0 ** torch.floor(torch.log2(7.5 / wmax.clamp_min(1e-300))), torch.ones_like(wmax))
    # ... lowered until no lo code saturates: one halving when the hi values are normal f16 numbers (|lo| <= 2^-11 |hi|), more
    # when they are f16 subnormals (round 4: the packers iterate; they used to halve at most once)
    lmax = (Wl_g.abs() * rho).amax(dim=1)
    cap = torch.where(lmax > 0, 2.0 ** torch.floor(torch.log2(7.5 / lmax.clamp_min(1e-300))), torch.full_like(lmax, float("inf")))
    return torch.minimum(a2, cap)


def x2_matmul(x, W, x_scale_hi=4.0, rho=4096.0, w_target=8192.0):
    """x [..., K] (float), W [N, K] -> x @ W.T evaluated in the x2 arithmetic (float64 accumulation stands in for fp32)."""
    x, W = x.double(), W.double()
    K = W.shape[1]
    mx = float(W.abs().max())
    sc = 2.0 ** np.floor(np.log2(w_target / mx)) if mx > 0 else 1.0
    Ws = W * sc
    Wh = f16(Ws)
    Wl = Ws - Wh
    xh = f16(x)
    xl = x - xh
    y = xh @ Wh.t()
    Bh = q_e2m3(xh * x_scale_hi)
    Bl = q_e2m3(f16(xl * rho) * x_scale_hi)          # the lo plane travels as f16 (pre-scaled by rho) into the conversion
    for g in slot_groups(K):
        alpha = weight_alpha(Wh[:, g], Wl[:, g], rho)
        Ah = q_e2m3(Wh[:, g] * alpha[:, None])
        Al = q_e2m3(Wl[:, g] * rho * alpha[:, None])
        cross = (Bl[..., g] @ Ah.t() + Bh[..., g] @ Al.t()) / (alpha * x_scale_hi * rho)
        y = y + cross
    return (y / sc)


def x3_matmul(x, W, w_target=8192.0):
    """The same product in the three-product f16 arithmetic of the x3 engines (for comparison)."""
    x, W = x.double(), W.double()
    mx = float(W.abs().max())
    sc = 2.0 ** np.floor(np.log2(w_target / mx)) if mx > 0 else 1.0
    Ws = W * sc
    Wh = f16(Ws)
    Wl = f16(Ws - Wh)
    xh = f16(x)
    xl = f16(x - xh)
    return (xh @ Wh.t() + xl @ Wh.t() + xh @ Wl.t()) / sc


def x2_operands_matmul(x, Whi, groups, dynamic=True, x_scale_hi=4.0, rho=4096.0):
    """The kernel's arithmetic on DECODED operands: Whi [N, K] f16 values (feature order), groups = list of
    (feature index tensor [16], codes_hi [N, 16], codes_lo [N, 16], block_scale [N] = 1 / alpha) -> x @ W.T.
    dynamic: per-(row of x, group) power-of-two activation scale from the largest |x| of the group (synthesis engine);
    otherwise the static scale of the field engine (|x| <= 1)."""
    x = x.double()
    xh = f16(x)
    xl = f16((x - xh) * rho)                       # lo' as the f16 fragment the conversion reads
    y = xh @ Whi.t()
    for feats, a_hi, a_lo, s in groups:
        gh, gl = xh[..., feats], xl[..., feats]
        if dynamic:
            amax = x[..., feats].abs().amax(dim=-1, keepdim=True).float().double()      # the kernel's fp32 running maximum
            e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -113)))
            tight = amax / 2.0 ** e < 1.875                                               # then |hi| / 2^(e-2) < 7.5
            cs = torch.where(tight, 2.0 ** (e - 2), 2.0 ** (e - 1))
        else:
            cs = torch.full_like(gh[..., :1], 1.0 / x_scale_hi)
        Bh, Bl = q_e2m3(gh / cs), q_e2m3(gl / cs)
        y = y + (Bl @ a_hi.t() + Bh @ a_lo.t()) * s * (cs / rho)
    return y


def x2_weight_operands(W, rho=4096.0):
    """What SynthesisPlan.pack_stream_x2 encodes (no matrix scale): -> (Whi, groups) for x2_operands_matmul."""
    W = W.double()
    Wh = f16(W)
    Wl = W - Wh
    groups = []
    for g in slot_groups(W.shape[1]):
        alpha = weight_alpha(Wh[:, g], Wl[:, g], rho)
        groups.append((g, q_e2m3(Wh[:, g] * alpha[:, None]), q_e2m3(Wl[:, g] * rho * alpha[:, None]), 1.0 / alpha))
    return Wh, groups
