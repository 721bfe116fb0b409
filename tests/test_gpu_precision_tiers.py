"""The reduced-precision tiers of the x3t engines (BASELINE config 5's "fp16 MFMA path"): one or two f16 products per
operand pair instead of three.  They are OUTSIDE the 1e-3 parity budget by design; the tolerances stated here are theirs:

    field  "f16x1t"   plain f16 products in the hidden GEMMs           render (rgb) within 3e-2, features within 5e-2
    synth  "f16w2t"   weights f16 hi+lo, activations one f16 value     image within 1e-2 of the oracle
    synth  "f16x1t"   plain f16 products                               image within 1e-2 of the oracle

(measured on MI355X: the two-product synthesis variant is NOT more accurate than the single-product one -- the rounding of
the activations to one f16 value dominates both, 1e-3 .. 5e-3 -- so it buys nothing; it stays in the C ABI as the measured
evidence.)  Each tier must also be clearly LESS accurate than the default engine: a tier that silently ran the 3-product
kernel would pass its tolerance but fail that check."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import load_golden, rel_err_channels

pytestmark = pytest.mark.gpu
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")
DEV = "cuda"


def _gen(width):
    meta = dict(load_golden("gen_tiny_mixed")["meta"])
    meta.update(hidden_dim=width, latent_dim=width, feature_dim=width, gen_height=64, gen_width=32, render_height=16,
                render_width=8, num_steps=32)
    meta["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(width)
    G = gens.Map3DGenerator(**meta).to(DEV).eval()
    G.set_device(DEV)
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    cond = synthetic.make_conditions(2, n_vertices=500, seed=5)
    z = torch.randn(2, width)
    jit = torch.rand(2, 16 * 8, 32, 1)
    ref = O.generator_forward(sd, {k: v for k, v in meta.items() if k != "neural_field_cls"}, z, cond, jit, None)
    run = lambda: G.forward(z.to(DEV), {k: v.to(DEV) for k, v in cond.items()}, jitter=jit.to(DEV), **meta)
    return G, run, ref


@pytest.mark.parametrize("width", [256, 384])
def test_field_single_product_tier(width):
    G, run, ref = _gen(width)
    G.neural_field.precision = "f16x3t"
    G.synthesis_plan(DEV).engine = "bf16x3t"
    e3 = rel_err_channels(run()["rgbs_render"].cpu(), ref["rgbs_render"])
    G.neural_field.precision = "f16x1t"
    e1 = rel_err_channels(run()["rgbs_render"].cpu(), ref["rgbs_render"])
    print(f"field width {width}: render error x3t {e3:.2e}, x1t {e1:.2e}")
    assert e3 < 1e-5
    assert 5 * e3 < e1 < 3e-2          # a real single-product result: clearly worse than the split engine, inside its tolerance


@pytest.mark.parametrize("width", [256, 420])
def test_synthesis_reduced_product_tiers(width):
    G, run, ref = _gen(width)
    G.neural_field.precision = "f16x3t"
    errs = {}
    for eng in ("bf16x3t", "f16w2t", "f16x1t"):
        G.synthesis_plan(DEV).engine = eng
        errs[eng] = rel_err_channels(run()["rgbs"].cpu(), ref["rgbs"])
    print(f"synthesis width {width}: image error " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["bf16x3t"] < 2e-4
    assert 5 * errs["bf16x3t"] < errs["f16w2t"] < 1e-2
    assert 5 * errs["bf16x3t"] < errs["f16x1t"] < 1e-2


def test_tiers_are_opt_in():
    G, _, _ = _gen(256)
    assert G.neural_field.precision == "f16x2" and G.synthesis_plan(DEV).engine == "f16x2"      # inside the 1e-3 budget
