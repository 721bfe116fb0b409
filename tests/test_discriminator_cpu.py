"""UNetDiscriminator, the discriminator-step losses and the R1 penalty against vectors captured from the reference
(tests/golden/make_golden_train.py: the reference's UNetDiscriminator and PhaseTrainer loss methods on the same weights /
inputs).  The discriminator runs through torch, so these are CPU tests."""
import importlib
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err

disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
losses = importlib.import_module("3dhumangan_amd.lib.trainers.losses")
trainers = importlib.import_module("3dhumangan_amd.lib.trainers")


@pytest.fixture(scope="module")
def fx():
    g = load_golden("disc_tiny")
    info = json.load(open(os.path.join(GOLDEN, "disc_tiny.json")))
    D = disc.UNetDiscriminator(**info["kwargs"]).eval()
    D.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in g["state"].items()}, strict=True)
    return g, info, D


def test_state_dict_schema_and_forward(fx):
    g, info, D = fx
    assert set(D.state_dict()) == set(g["state"])
    for name, x in (("out_real", g["real"]), ("out_fake", g["fake"])):
        out = D(x, None, 1.0)
        assert set(out) == set(g[name])
        for k, v in out.items():
            assert v.shape == g[name][k].shape
            # fp32 rounding order only: round 4 pools the SUM of the two branches of a down block once (ResBlock.forward)
            assert rel_err(v.detach(), g[name][k]) < 3e-6, (name, k)


def test_losses_match_the_reference(fx):
    g, info, D = fx
    meta = info["meta"]
    real = g["real"].clone().requires_grad_(True)
    out_real, out_fake = D(real, None, 1.0), D(g["fake"], None, 1.0)
    gan = losses.logistic_d_loss(out_real["prediction"], out_fake["prediction"], meta["gan_lambda"])
    assert rel_err(gan.detach(), g["loss"]["gan"]) < 1e-6
    grad = losses.r1_gradient(real, out_real, meta["gan_lambda"])
    # the reference's penalty: channels of sample 0 (see lib/trainers/losses.py) ...
    assert rel_err(losses.r1_penalty(grad, meta["r1_lambda"], "reference").detach(), g["loss"]["r1"]) < 1e-6
    # ... which is NOT the per-sample mean
    per_sample = losses.r1_penalty(grad, meta["r1_lambda"], "per_sample")
    assert abs(float(per_sample) / float(g["loss"]["r1"]) - 1) > 1e-3
    assert rel_err(losses.r1_statistic(grad, "per_sample").detach(), grad.detach().flatten(1).pow(2).sum(1)) < 1e-7
    s_real, acc, p_real = losses.segmentation_loss(out_real["segments"], g["gt_segments"], meta["label_dim"])
    s_gen, _, p_gen = losses.segmentation_loss(out_fake["segments"], torch.zeros_like(g["gt_segments"]), meta["label_dim"])
    for a, b in ((s_real, "seg_real"), (s_gen, "seg_gen"), (acc, "acc_real"), (p_real, "prob_real"), (p_gen, "prob_gen")):
        assert rel_err(a.detach(), g["loss"][b]) < 1e-6, b


def test_discriminator_step_gradient_matches_the_reference(fx):
    """The whole D loss (logistic + 4 x R1 + segmentation) and its gradient w.r.t. a conv weight -- i.e. the double
    backward through the spectral-norm convs -- as the reference computes them; then the optimiser step moves the weights."""
    g, info, D = fx
    meta = dict(info["meta"])
    D = disc.UNetDiscriminator(**info["kwargs"]).eval()
    D.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in g["state"].items()}, strict=True)
    opt = torch.optim.SGD(D.parameters(), lr=0.0)                     # lr 0: the step leaves .grad in place for inspection
    res = trainers.discriminator_step(D, opt, g["real"], g["fake"], g["gt_segments"], meta, do_r1=True, r1_mode="reference")
    assert rel_err(res["loss"], g["loss"]["total"]) < 1e-6
    key = info["grad_key"]
    got = dict(D.named_parameters())[key].grad
    assert rel_err(got, g["grad"][key]) < 1e-5
    before = dict(D.named_parameters())[key].detach().clone()
    trainers.discriminator_step(D, torch.optim.Adam(D.parameters(), lr=1e-3, betas=(0.0, 0.9)), g["real"], g["fake"], g["gt_segments"],
                                meta, do_r1=True, grad_clip=10.0)
    assert not torch.equal(before, dict(D.named_parameters())[key].detach())


def test_block_count_follows_the_image_size():
    D = disc.UNetDiscriminator(latent_dim=8, gen_height=16, gen_width=8, label_dim=2, discriminator_blocks=6)
    assert D.num_blocks == 3 and len(D.body_down) == 3 and len(D.body_up) == 3
    out = D(torch.zeros(1, 3, 16, 8), None, 1.0)
    assert out["prediction"].shape == (1, 1, 16, 8) and out["segments"].shape == (1, 2, 16, 8) and out["latents"].shape == (1, 8)
    D6 = disc.UNetDiscriminator(latent_dim=8, gen_height=16, gen_width=8, label_dim=2, semantic_dim=3, dual_discrimination=True)
    o6 = D6(torch.zeros(2, 6, 16, 8), None, 1.0)
    assert o6["semantics"].shape == (2, 3, 16, 8) and o6["segments"].shape == (2, 2, 16, 8)
