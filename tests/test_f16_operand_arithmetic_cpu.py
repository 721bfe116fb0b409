"""CPU restatement of the AMP tier's matrix arithmetic (csrc/conv_x3.hip HALF, csrc/wgrad_x3.hip HALF; reference: nn.Conv2d / nn.Linear
inside torch.cuda.amp.autocast, lib/trainers/base_trainer.py:50-51).

The activation (and its gradient) is an f16 tensor: exact in ONE plane.  The kernels therefore multiply the f16 values themselves:
  * convolution / dense forward and data gradient:  y = W_hi x + W_lo x  with W_hi = f16(W), W_lo = f16(W - W_hi)  (two F16 matrix
    products, fp32 accumulation) -- the weight enters to max(2^-22 |W|, 2^-25) (the lo plane lives in f16's subnormal range for
    |W| < 2^-3), where autocast's own f16 GEMM rounds it to 11 bits;
  * weight gradient: dW = dY^T X, both operands f16: ONE product, every term exact in fp32 before the accumulation.
This file checks those statements in numpy (float32 accumulation emulated by float64 sums of exactly representable products)."""
import numpy as np

rng = np.random.default_rng(0)


def f16(a):
    return a.astype(np.float16).astype(np.float64)


def test_two_f16_weight_planes_carry_the_weight_to_2_pow_minus_25():
    w = rng.normal(0, 0.05, 20000)
    hi = f16(w)
    lo = f16(w - hi)
    assert (np.abs(hi - w) / np.abs(w)).max() > 2.0 ** -12.5                 # one plane: 11 bits (what autocast's f16 GEMM sees)
    assert (np.abs(hi + lo - w) <= np.maximum(2.0 ** -22 * np.abs(w), 2.0 ** -25)).all()
    big = rng.normal(0, 1.0, 20000)
    big = big[np.abs(big) > 0.125]
    assert (np.abs(f16(big) + f16(big - f16(big)) - big) / np.abs(big)).max() < 2.0 ** -21     # ordinary magnitudes: 22 bits


def test_products_of_f16_values_are_exact_in_fp32():
    a = rng.normal(0, 1, 4096).astype(np.float16)
    b = rng.normal(0, 1, 4096).astype(np.float16)
    p32 = a.astype(np.float32) * b.astype(np.float32)    # 11 x 11 significant bits fit the 24 of fp32
    assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def test_conv_as_two_products_beats_the_rounded_weight_gemm():
    M, K, N = 64, 1152, 32                                # 64 pixels, 128 channels x 9 taps, 32 output channels
    x = rng.normal(0, 1, (M, K)).astype(np.float16).astype(np.float64)
    w = rng.normal(0, 1 / np.sqrt(K), (N, K))
    exact = x @ w.T
    hi = f16(w)
    lo = f16(w - hi)
    ours = (x @ hi.T + x @ lo.T).astype(np.float32)       # fp32 accumulation of exact products (sums emulated in float64)
    library = (x @ hi.T).astype(np.float32)               # autocast: the weight rounded to f16 once
    scale = np.abs(exact).max()
    assert np.abs(ours - exact).max() / scale < 2e-6
    assert np.abs(library - exact).max() / scale > 5e-5   # the 11-bit weight is what limits the library path


def test_bounded_operand_split_gives_the_bits_of_the_general_one():
    """csrc/x3_common.hpp: split2_x2_bounded (round 5, the x2 field's sine outputs) computes the scaled residual plane as ONE mixed
    FMA per value, f16(a * 2^12 - f16(hi * 2^12)), where the general split2_x2 computes f16((a - hi) * 2^12) in two steps.  Both round
    the SAME exact real number once, as long as hi * 2^12 is an f16 without rounding: |hi| < 16 (no overflow) -- f16 subnormals of
    hi become normal numbers, so nothing is lost at the small end either.  Restated in float64 (every intermediate below is exact in
    float64: 24-bit a, 11-bit hi, powers of two) over random and edge-case float32 inputs; outside the range the two differ, which
    is why the synthesis engine (activations up to 2^15) keeps the general split."""
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(0)
    a = torch.cat([torch.rand(200000, generator=g) * 2 - 1,                                   # the sine's range
                   (torch.rand(50000, generator=g) * 2 - 1) * 15.99,                           # up to the limit
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -14, 2.0 ** -24, 3e-8, 6.1e-5, 0.99999994, 15.99, -15.99,
                                 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -12, 0.5 + 2.0 ** -13]),
                   torch.from_numpy(np.ldexp(np.random.default_rng(1).uniform(1, 2, 20000), np.random.default_rng(2).integers(-30, 0, 20000))).float()])
    a64 = a.double()
    hi = a.to(torch.float16)                                     # v_cvt_pk_f16_f32 (round to nearest even)
    hi64 = hi.double()
    general = ((a64 - hi64) * 4096.0).to(torch.float16)          # v_fma_mix_f32 (exact residual), then v_fma_mixlo/hi_f16 (* 2^12, one rounding)
    hs = (hi64 * 4096.0).to(torch.float16)                       # v_pk_mul_f16 by 4096
    assert torch.equal(hs.double(), hi64 * 4096.0)               # ... which is exact in this range
    bounded = (a64 * 4096.0 - hs.double()).to(torch.float16)     # v_fma_mix{lo,hi}_f16 a, 4096, -hs: exact sum, one rounding
    assert torch.equal(general.view(torch.int16), bounded.view(torch.int16))
    assert torch.isfinite(general.float()).all()
    # outside the range the bounded form breaks (hi * 2^12 overflows f16): the reason it is used for bounded activations only
    big = torch.tensor([16.0, 100.0, -3000.0]).double()
    assert not torch.isfinite((big.to(torch.float16).double() * 4096.0).to(torch.float16).float()).any()
