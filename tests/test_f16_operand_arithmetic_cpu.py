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
