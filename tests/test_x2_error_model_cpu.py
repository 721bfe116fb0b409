"""CPU error model of the x2 arithmetic (tests/x2_emulation.py: one f16 product + one block-scaled fp6 product for the two
cross terms) on the REFERENCE's own field vectors at MAP3DBN512's width: what csrc/field_x2.hip must reproduce on the GPU,
and the evidence that the scheme sits well inside the 1e-3 budget before any kernel runs."""
import pytest
import torch

from conftest import load_golden, rel_err
from x2_emulation import q_e2m3, x2_matmul, x3_matmul


def _field(state, points, freq, phase, geo, dirs, input_scaler, mm):
    """lib/implicit_funcitions/modulated.py:41-75 with the hidden-layer contractions routed through `mm` (float64 elsewhere)."""
    p = "neural_field."
    st = {k: v.double() for k, v in state.items()}
    hd = st[p + "sigma_layer.weight"].shape[1]

    def lin(name, x, hidden=False, cols=None):
        W, b = st[p + name + ".weight"], st[p + name + ".bias"]
        if cols is not None:
            W = W[:, cols]
        return (mm(x, W) if hidden else x @ W.t()) + b

    f = (freq.double() * 15 + 30).unsqueeze(1)
    ph = phase.double().unsqueeze(1)
    a = torch.sin(30.0 * lin("first_layer_coord.layer", points.double() * input_scaler))
    g = torch.sin(30.0 * lin("first_layer_mod.layer", geo.double()))
    x = torch.cat([a, g], dim=-1)
    for k in range(4):
        sl = slice(k * hd, (k + 1) * hd)
        x = torch.sin(f[..., sl] * lin(f"network.{k}.layer", x, hidden=True) + ph[..., sl])
    sigma = lin("sigma_layer", x)
    Wc, bc = st[p + "color_layer_sine.layer.weight"], st[p + "color_layer_sine.layer.bias"]
    c = mm(x, Wc[:, 3:]) + dirs.double() @ Wc[:, :3].t() + bc
    c = torch.sin(f[..., -hd:] * c + ph[..., -hd:])
    rgb = torch.sigmoid(lin("color_layer_linear", c))
    feat = lin("feature_layer_linear", c, hidden=True)
    return torch.cat([rgb, feat, sigma], dim=-1)


def test_e2m3_quantiser():
    v = torch.tensor([0.0, 0.06, 0.0625, 0.19, 0.9375, 1.06, 1.0625, 1.9, 2.1, 2.125, 3.9, 4.3, 7.4, 9.0, -0.3, -7.6], dtype=torch.float64)
    want = torch.tensor([0.0, 0.0, 0.0, 0.25, 1.0, 1.0, 1.0, 1.875, 2.0, 2.0, 4.0, 4.5, 7.5, 7.5, -0.25, -7.5], dtype=torch.float64)
    assert torch.equal(q_e2m3(v), want)                       # nearest-even ties (0.0625 -> 0, 0.9375 -> 1, 1.0625 -> 1, 2.125 -> 2), saturation


@pytest.mark.parametrize("hidden", [256])
def test_x2_field_error_on_reference_vectors(hidden):
    g = load_golden(f"field_h{hidden}")
    state = {k: v.float() for k, v in g["state"].items()}
    args = (state, g["points"], g["freq"], g["phase"], g["geo"], g["dirs"], 2.0 / 2.85)
    exact = _field(*args, mm=lambda x, W: x @ W.t())
    assert rel_err(exact, g["out"]) < 2e-5                    # the float64 restatement itself reproduces the reference
    x2 = _field(*args, mm=x2_matmul)
    x3 = _field(*args, mm=x3_matmul)
    sls = {"rgb": slice(0, 3), "feat": slice(3, 3 + hidden), "sigma": slice(3 + hidden, 4 + hidden)}
    e2 = {k: rel_err(x2[..., s], exact[..., s]) for k, s in sls.items()}
    e3 = {k: rel_err(x3[..., s], exact[..., s]) for k, s in sls.items()}
    print("x2 arithmetic vs float64:", e2, " x3:", e3)
    # budget: 1e-3 at the operator boundary (tests/test_gpu_field.py: FIELD_TOL); the model predicts ~1e-4 at worst
    assert max(e2.values()) < 3e-4, e2
    assert max(e3.values()) < 3e-5, e3
