"""The x2 synthesis engine's range guard and its behaviour on hostile / trained-like weights ("make the x2 default
safe").  The x2 arithmetic carries activations as f16 planes (hi = f16(x), f16(lo * 2^12)): finite only for |x| < 2^15.  The
engine raises a device flag when it leaves that range and the bf16 engine recomputes the image (SynthesisPlan.run); weights
outside the range make the plan decline x2 when it is built.  Reference semantics: lib/components/map3d_layers.py:193-238
(fp32 convolutions: no range limit)."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import load_golden, rel_err, rel_err_channels

pytestmark = pytest.mark.gpu
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
DEV = "cuda"


def make(width, gh, gw, rh, rw, seed, mutate=None, mode="mixed"):
    meta = dict(load_golden("gen_tiny_mixed")["meta"])
    meta.update(hidden_dim=width, latent_dim=width, feature_dim=width, gen_height=gh, gen_width=gw, render_height=rh,
                render_width=rw, num_steps=8, map3d_mode=mode)
    meta["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(seed)
    G = gens.Map3DGenerator(**meta).eval()
    if mutate is not None:
        with torch.no_grad():
            mutate(G)
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    G = G.to(DEV)
    G.set_device(DEV)
    return G, meta, sd


def oracle_rgb(sd, meta, fmap, style):
    """fmap [B, Hr*Wr, C] channels-last, style [B, C] -> the oracle's synthesis network on the resized maps."""
    B, C = style.shape
    H, W, Hr, Wr = meta["gen_height"], meta["gen_width"], meta["render_height"], meta["render_width"]
    fm = fmap.view(B, Hr, Wr, C).permute(0, 3, 1, 2)
    up = torch.nn.functional.interpolate(fm, (H, W), mode="bilinear")
    x0 = O.synthesis_input(sd, B, H, W)
    return O.synthesis_network(sd, x0, up, style.view(B, 1, C), meta["map3d_mode"], tuple(meta["mod_blocks"]),
                               meta["synthesis_blocks"])["final"]


def run(G, meta, fmap, style):
    return G._synthesize(fmap.to(DEV), style.to(DEV), (meta["render_height"], meta["render_width"]))


@pytest.mark.parametrize("width", [64, 256])
def test_guard_is_quiet_in_range_and_changes_nothing(width):
    G, meta, sd = make(width, 64, 64, 12, 12, seed=1)
    plan = G.synthesis_plan(DEV)
    assert plan.engine == "f16x2" and plan.x2_guard
    plan.x2_monitor = False                             # the RANGE guard is under test here (the sampled error monitor: test_gpu_x2_monitor.py)
    B = 2
    fmap, style = torch.randn(B, 144, width), torch.randn(B, width)
    out = run(G, meta, fmap, style)
    assert not plan.x2_fell_back()
    plan.x2_guard = False
    plain = run(G, meta, fmap, style)
    assert torch.equal(out, plain)                      # the guarded launch is the same kernel; the x3 launch returned at once
    assert rel_err(out.cpu(), oracle_rgb(sd, meta, fmap, style)) < 5e-4


@pytest.mark.parametrize("width", [64, 256])
def test_activations_beyond_the_f16_planes_are_recomputed_on_bf16(width):
    """Feature maps scaled so that the shared-MLP activations reach ~1e5: the x2 engine alone returns garbage (non-finite or
    far off), the guarded pair returns the oracle's image."""
    G, meta, sd = make(width, 64, 64, 12, 12, seed=2)
    plan = G.synthesis_plan(DEV)
    assert plan.engine == "f16x2"
    plan.x2_monitor = False                             # the RANGE guard alone
    B = 2
    fmap, style = torch.randn(B, 144, width) * 6e4, torch.randn(B, width)
    ref = oracle_rgb(sd, meta, fmap, style)
    assert torch.isfinite(ref).all()
    out = run(G, meta, fmap, style)
    assert plan.x2_fell_back() and plan.x2_fallback_items() == [0, 1]
    assert torch.isfinite(out).all() and rel_err(out.cpu(), ref) < 1e-3
    # one item out of range, one in range: only the first is redone, the second keeps its x2 pixels (round 6: per-item flags)
    mixed = torch.cat([fmap[:1], torch.randn(1, 144, width)])
    out = run(G, meta, mixed, style)
    assert plan.x2_fallback_items() == [0]
    plan.x2_guard = False
    assert torch.equal(run(G, meta, mixed, style)[1], out[1])
    plan.x2_guard = True
    assert rel_err(out.cpu(), oracle_rgb(sd, meta, mixed, style)) < 1e-3
    plan.x2_guard = False
    raw = run(G, meta, fmap, style).cpu()
    assert (not torch.isfinite(raw).all()) or rel_err(raw, ref) > 1e-2, "the scenario no longer breaks the unguarded x2 engine"
    # ... and the next in-range call is clean again (the flag is per run)
    plan.x2_guard = True
    run(G, meta, torch.randn(B, 144, width), style)
    assert not plan.x2_fell_back()


def test_unnormalised_spectral_norm_state_gets_the_oracles_answer():
    """with the spectral-norm vectors at their random initial values sigma = u . W v is small and every block
    multiplies the activations by ~1e3 (1e26 at the last block).  Whatever the plan picks, the image must be the oracle's."""
    def mutate(G):
        g = torch.Generator().manual_seed(7)
        for n, b in G.named_buffers():
            if n.endswith("weight_u") or n.endswith("weight_v"):
                v = torch.randn(b.shape, generator=g)
                b.copy_(v / v.norm())
    width = 256
    G, meta, sd = make(width, 64, 64, 12, 12, seed=3, mutate=mutate)
    plan = G.synthesis_plan(DEV)
    B = 2
    fmap, style = torch.randn(B, 144, width), torch.randn(B, width)
    ref = oracle_rgb(sd, meta, fmap, style)
    out = run(G, meta, fmap, style).cpu()
    print(f"engine {plan.engine}; oracle max |rgb| {float(ref.abs().max()):.3e}; fell back: {plan.x2_fell_back()}")
    assert float(ref.abs().max()) > 1e6                 # the scenario is the blown-up one
    assert plan.engine != "f16x2" or plan.x2_fell_back()
    if torch.isfinite(ref).all():
        assert torch.isfinite(out).all() and rel_err(out, ref) < 1e-3
    else:
        assert not torch.isfinite(out).all()


def test_weights_outside_the_f16_range_decline_x2_when_the_plan_is_built():
    def mutate(G):
        w = G.synthesis_network.network["m3d_5"].conv_0.weight_orig
        w.mul_(1e6)
        G.synthesis_network.network["m3d_5"].conv_0.weight_u.mul_(1e-6)      # sigma unchanged: the effective weight is 1e6 x
    G, meta, sd = make(64, 64, 64, 12, 12, seed=4, mutate=mutate)
    plan = G.synthesis_plan(DEV)
    assert plan.engine == "bf16x3" and not plan.x2_weights_in_range()
    fmap, style = torch.randn(1, 144, 64), torch.randn(1, 64)
    ref = oracle_rgb(sd, meta, fmap, style)
    out = run(G, meta, fmap, style).cpu()
    assert torch.isfinite(out).all() and rel_err(out, ref) < 1e-3


def trained_like(width, meta):
    """-> mutate(G): statistics a trained checkpoint has and a fresh one has not -- per-channel conv gains spread over 2^12,
    affine BatchNorm weights / biases away from (1, 0), heavy-tailed gamma / beta matrices -- and then, as training leaves them,
    BatchNorm running statistics that MATCH the activations (far from (0, 1)) and converged spectral-norm vectors: 30 train-mode
    passes of the oracle over a calibration batch (momentum 0.1)."""
    def mutate(G):
        g = torch.Generator().manual_seed(11)
        for name, m in G.synthesis_network.network.items():
            for s in range(2):
                conv, sp = getattr(m, f"conv_{s}"), getattr(m, f"spade_{s}")
                w = conv.weight_orig
                w.mul_(torch.exp2(torch.rand(w.shape[0], generator=g) * 12.0 - 6.0).view(-1, 1, 1, 1))      # 2^-6 .. 2^6 per channel
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.3)
                bn = sp.first_norm
                bn.weight.copy_(torch.exp2(torch.randn(bn.weight.shape, generator=g)))
                bn.bias.copy_(torch.randn(bn.bias.shape, generator=g))
                for lin in (sp.mlp_gamma, sp.mlp_beta):
                    t = torch.distributions.StudentT(3.0).sample(lin.weight.shape)                         # heavy tails
                    lin.weight.copy_(t * lin.weight.std())
        sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
        H, W, Hr, Wr = meta["gen_height"], meta["gen_width"], meta["render_height"], meta["render_width"]
        B = 2
        for it in range(30):
            fm = torch.randn(B, width, Hr, Wr, generator=g)
            up = torch.nn.functional.interpolate(fm, (H, W), mode="bilinear")
            buf = {}
            O.synthesis_network(sd, O.synthesis_input(sd, B, H, W), up, torch.randn(B, 1, width, generator=g), meta["map3d_mode"],
                                tuple(meta["mod_blocks"]), meta["synthesis_blocks"], training=True, buffers_out=buf)
            sd.update({k: v.detach() for k, v in buf.items()})
        G.load_state_dict(sd, strict=True)
    return mutate


@pytest.mark.parametrize("width", [64, 256])
def test_trained_like_weights_stay_inside_half_the_budget_or_are_declined(width):
    """Worst per-channel error of the default engine on trained-like statistics: < 5e-4 of the channel's range (half the 1e-3
    budget) -- or the image came from the bf16 engine."""
    meta0 = dict(load_golden("gen_tiny_mixed")["meta"])
    meta0.update(gen_height=64, gen_width=64, render_height=12, render_width=12)
    G, meta, sd = make(width, 64, 64, 12, 12, seed=5, mutate=trained_like(width, meta0))
    plan = G.synthesis_plan(DEV)
    B = 4
    fmap, style = torch.randn(B, 144, width), torch.randn(B, width)
    ref = oracle_rgb(sd, meta, fmap, style)
    assert torch.isfinite(ref).all()
    out = run(G, meta, fmap, style).cpu()
    fell = plan.engine == "f16x2" and plan.x2_fell_back()
    e_img, e_ch = rel_err(out, ref), rel_err_channels(out, ref)
    plan.engine = "bf16x3"
    e3 = rel_err_channels(run(G, meta, fmap, style).cpu(), ref)
    rm = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_mean") and "synthesis" in k])
    rv = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_var") and "synthesis" in k])
    print(f"width {width}: fell back {fell}; default image err {e_img:.2e}, worst channel {e_ch:.2e}; bf16x3 worst channel "
          f"{e3:.2e}; max |rgb| {float(ref.abs().max()):.3e}; running_mean in [{float(rm.min()):.2g}, {float(rm.max()):.2g}], "
          f"running_var in [{float(rv.min()):.2g}, {float(rv.max()):.2g}]")
    assert e3 < 1e-3
    assert fell or e_ch < 5e-4


def test_wide_engine_x2_tier_is_guarded_too():
    """The LDS-resident engine's x2 tier (f16x2t: MAP3DBN 384, MAP3DBN512L 420) carries the same flag: in range it changes
    nothing, out of range the bf16 tier recomputes the image."""
    width = 300
    G, meta, sd = make(width, 40, 24, 9, 5, seed=6, mode="isolated")
    plan = G.synthesis_plan(DEV)
    assert plan.engine == "f16x2t"
    fmap, style = torch.randn(2, 45, width), torch.randn(2, width)
    out = run(G, meta, fmap, style)
    assert not plan.x2_fell_back()
    assert rel_err(out.cpu(), oracle_rgb(sd, meta, fmap, style)) < 1e-3
    plan.x2_guard = False
    assert torch.equal(out, run(G, meta, fmap, style))
    plan.x2_guard = True
    big = fmap * 6e4
    ref = oracle_rgb(sd, meta, big, style)
    assert torch.isfinite(ref).all()
    out = run(G, meta, big, style)
    assert plan.x2_fell_back()
    assert torch.isfinite(out).all() and rel_err(out.cpu(), ref) < 1e-3
