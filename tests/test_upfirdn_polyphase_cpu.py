"""The index algebra of the polyphase upfirdn2d kernel (csrc/upfirdn2d.hip: struct Poly, upfirdn2d_poly) restated and checked on
the CPU against the definition of upfirdn2d, for every compiled geometry and every padding phase: a 1-D signal is resampled through
the kernel's tile / window / live-tap arithmetic (LDS tile origin, per-thread window start, register index of each live tap) and
must equal pad -> zero-stuff -> correlate -> decimate.  2-D is the same arithmetic per axis."""
import numpy as np
import pytest

VX, TW = 4, 64                             # patch and tile of the x axis (the y axis uses VY = 2 or 4, TH = 16 VY: same formulas)


def floor_div(a, b):
    return a // b                          # Python floors


def reference(x, f_corr, up, down, pad0, out_n):
    """out[o] = sum_k f_corr[k] * padded_upsampled[o * down + k], zero outside."""
    n = len(x)
    out = np.zeros(out_n)
    for o in range(out_n):
        for k in range(len(f_corr)):
            u = o * down + k - pad0
            if u % up == 0 and 0 <= u // up < n:
                out[o] += f_corr[k] * x[u // up]
    return out


def poly_axis(x, f_corr, U, D, pad0, out_n, V, T):
    F = len(f_corr)
    R = (-pad0) % U
    assert (V * D) % U == 0 and T % V == 0
    W = (R + (V - 1) * D + F - 1) // U + 1                 # register window of a thread
    IN = (R + (T - 1) * D + F - 1) // U + 1                # LDS tile of a workgroup
    n = len(x)
    out = np.zeros(out_n)
    for o0 in range(0, out_n, T):                          # workgroup
        i0 = floor_div(o0 * D - pad0, U)
        tile = np.array([x[i0 + l] if 0 <= i0 + l < n else 0.0 for l in range(IN)])
        for tg in range(T // V):                           # thread
            ob = o0 + V * tg
            if ob >= out_n:
                break
            start = V * tg * D // U
            assert start + W <= IN, "window inside the tile"
            win = tile[start:start + W]
            # the kernel's claim: base coordinate of the patch = (i0 + start) * U + R
            assert ob * D - pad0 == (i0 + start) * U + R
            for v in range(V):
                if ob + v >= out_n:
                    continue
                acc = 0.0
                for k in range(F):
                    if (R + v * D + k) % U:
                        continue                           # tap lands between samples of the zero-stuffed signal
                    acc += win[(R + v * D + k) // U] * f_corr[k]
                out[ob + v] = acc
    return out


@pytest.mark.parametrize("U,D,F", [(2, 1, 4), (1, 2, 4), (1, 1, 4), (2, 1, 1), (1, 2, 1), (1, 1, 1)])
@pytest.mark.parametrize("V,T", [(4, 64), (2, 32)])
def test_polyphase_axis_matches_the_definition(U, D, F, V, T):
    rng = np.random.default_rng(U * 10 + D + F)
    for n in (5, 37, 64, 70, 131):
        x = rng.standard_normal(n)
        f = rng.standard_normal(F)
        for pad0 in range(-3, 6):
            for pad1 in (0, 1, 3):
                out_n = (n * U + pad0 + pad1 - F + D) // D
                if out_n < 1:
                    continue
                ref = reference(x, f, U, D, pad0, out_n)
                got = poly_axis(x, f, U, D, pad0, out_n, V, T)
                assert np.allclose(got, ref, atol=1e-12), (n, pad0, pad1)


def test_live_taps_of_the_stylegan_upsampler():
    """2x up with a 4-tap filter: every output uses exactly 2 of the 4 taps per axis (the polyphase saving), for both phases."""
    for R in (0, 1):
        for v in range(VX):
            live = [k for k in range(4) if (R + v + k) % 2 == 0]
            assert len(live) == 2
    # window sizes quoted in the kernel's header comment: at most 4 x 4 input samples for a 4 x 4 patch at 2x up
    assert max((R + 3 + 3) // 2 + 1 for R in (0, 1)) == 4
