"""BASELINE config 4 (the adversarial training iteration, D step with R1 + G step) on the GPU.

  * UNetDiscriminator forward, the D-step losses and the weight gradient of the whole D loss (the double backward of the R1
    penalty through the spectral-norm convolutions) against the reference's vectors (tests/golden/disc_tiny.npz), on MI355X;
  * generator_step's loss algebra and gradients against PhaseTrainer._train_generator (tests/golden/gstep_tiny.npz), on MI355X;
  * two whole iterations at config 4's per-GPU geometry (MAP3DBN512, 512x256 image, 96x48 rays x 32 samples, batch 4): finite
    losses, every parameter that receives a gradient moves, the discriminator treats the samples of a batch independently, and
    the differentiable generator path in eval mode reproduces the parity-tested inference engines at this geometry.
"""
import importlib
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err, rel_err_channels

pytestmark = pytest.mark.gpu
DEV = "cuda"
disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
losses = importlib.import_module("3dhumangan_amd.lib.trainers.losses")
trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
configs = importlib.import_module("3dhumangan_amd.configs")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")
ema_mod = importlib.import_module("3dhumangan_amd.lib.components.ema")
TOL = 1e-3


def _disc_fixture():
    g = load_golden("disc_tiny")
    info = json.load(open(os.path.join(GOLDEN, "disc_tiny.json")))
    D = disc.UNetDiscriminator(**info["kwargs"]).eval()
    D.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in g["state"].items()}, strict=True)
    return g, info, D.to(DEV)


def test_discriminator_forward_losses_and_r1_gradient_on_gpu():
    g, info, D = _disc_fixture()
    meta = dict(info["meta"])
    real, fake, gt = g["real"].to(DEV), g["fake"].to(DEV), g["gt_segments"].to(DEV)
    for name, x in (("out_real", real), ("out_fake", fake)):
        out = D(x, None, 1.0)
        for k, v in out.items():
            assert rel_err(v.detach().cpu(), g[name][k]) < TOL, (name, k)
    rq = real.clone().requires_grad_(True)
    out_real, out_fake = D(rq, None, 1.0), D(fake, None, 1.0)
    gan = losses.logistic_d_loss(out_real["prediction"], out_fake["prediction"], meta["gan_lambda"])
    assert rel_err(gan.detach().cpu(), g["loss"]["gan"]) < TOL
    grad = losses.r1_gradient(rq, out_real, meta["gan_lambda"])
    assert rel_err(losses.r1_penalty(grad, meta["r1_lambda"], "reference").detach().cpu(), g["loss"]["r1"]) < TOL
    s_real, acc, p_real = losses.segmentation_loss(out_real["segments"], gt, meta["label_dim"])
    s_gen, _, p_gen = losses.segmentation_loss(out_fake["segments"], torch.zeros_like(gt), meta["label_dim"])
    for a, b in ((s_real, "seg_real"), (s_gen, "seg_gen"), (acc, "acc_real"), (p_real, "prob_real"), (p_gen, "prob_gen")):
        assert rel_err(a.detach().cpu(), g["loss"][b]) < TOL, b
    # the whole D step: loss and the gradient of a conv weight (through the R1 double backward), as the reference computes them
    opt = torch.optim.SGD(D.parameters(), lr=0.0)
    res = trainers.discriminator_step(D, opt, real, fake, gt, meta, do_r1=True)          # default R1 statistic = the reference's
    assert rel_err(res["loss"].cpu(), g["loss"]["total"]) < TOL
    key = info["grad_key"]
    assert rel_err(dict(D.named_parameters())[key].grad.cpu(), g["grad"][key]) < TOL
    before = dict(D.named_parameters())[key].detach().clone()
    trainers.discriminator_step(D, torch.optim.Adam(D.parameters(), lr=1e-3, betas=(0.0, 0.9)), real, fake, gt, meta, do_r1=True,
                                grad_clip=10.0)
    assert not torch.equal(before, dict(D.named_parameters())[key].detach())


class _StubG(torch.nn.Module):
    """The stand-in generator the golden vector was captured with (tests/test_gstep_cpu.py)."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(16, 3 * 32 * 16)
        self.latent_pool = torch.nn.Embedding(4, 16)

    def forward(self, z, conditions, disable_synthesis=False, latent_indices=None, **kw):
        img = torch.tanh(self.lin(z)).view(z.shape[0], 3, 32, 16)
        return {"rgbs": img, "rgbs_render": img[:, :, ::4, ::4]}


def test_generator_step_on_gpu_matches_train_generator():
    info, g = json.load(open(os.path.join(GOLDEN, "gstep_tiny.json"))), load_golden("gstep_tiny")
    G = _StubG()
    G.load_state_dict(g["stub"])
    G = G.to(DEV)
    D = disc.UNetDiscriminator(**info["disc_kwargs"]).eval()
    D.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["disc"].items()})
    D = D.to(DEV)
    res = trainers.generator_step(G, D, torch.optim.SGD(G.parameters(), lr=0.0), g["z"].to(DEV), {}, dict(info["meta"]),
                                  gt_segments=g["data"]["rasterized_segments"].to(DEV), d_step_count=info["d_step"])
    assert res["topk"] == int(g["topk"])
    assert abs(float(res["loss"]) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    for n, p in G.named_parameters():
        if n in g["grad"]:
            assert rel_err(p.grad.cpu(), g["grad"][n]) < TOL, n
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n


def _cfg4(batch):
    cfg = {k: v for k, v in configs.MAP3DBN512.items() if isinstance(k, str)}
    cfg.update(gen_height=512, gen_width=256, render_height=96, render_width=48, num_steps=32, dataset_length=4, nerf_noise=0,
               last_back=cfg["eval_last_back"])
    torch.manual_seed(1234)
    G = gens.Map3DGenerator(**dict(cfg, neural_field_cls=impl.COORDCONCATSIREN)).to(DEV)
    G.set_device(DEV)
    gen = torch.Generator().manual_seed(5)
    cond = {k: v.to(DEV) for k, v in synthetic.make_conditions(batch, 6890, seed=5).items()}
    z = torch.randn(batch, cfg["latent_dim"], generator=gen).to(DEV)
    jitter = torch.rand(batch, 96 * 48, 32, 1, generator=gen).to(DEV)
    real = torch.randn(batch, 3, 512, 256, generator=gen).clamp(-1, 1).to(DEV)
    gt = torch.randint(0, cfg["label_dim"], (batch, 512, 256), generator=gen).to(DEV)
    return G, cfg, z, cond, jitter, real, gt


def test_config4_two_iterations_at_the_per_gpu_geometry():
    B = 4
    G, cfg, z, cond, jitter, real, gt = _cfg4(B)
    G.train()
    torch.manual_seed(99)
    D = disc.UNetDiscriminator(**cfg).to(DEV)
    meta = dict(cfg)
    meta.update(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, gen_lr=5e-5, betas=(0.0, 0.9))
    opt_d = torch.optim.Adam(D.parameters(), lr=cfg.get("disc_lr", 2e-4), betas=(0.0, 0.9))
    opt_g = trainers.make_generator_optimizer(G, meta)
    ema = ema_mod.ExponentialMovingAverage(G.parameters(), decay=0.999)
    g0 = {n: p.detach().clone() for n, p in G.named_parameters()}
    d0 = {n: p.detach().clone() for n, p in D.named_parameters()}
    fwd = {k: v for k, v in cfg.items() if isinstance(k, str)}
    for it in range(2):
        with torch.no_grad():
            fake = G(z, cond, jitter=jitter, **fwd)["rgbs"]
        assert fake.shape == (B, 3, 512, 256) and bool(torch.isfinite(fake).all())
        d = trainers.discriminator_step(D, opt_d, real, fake, gt, meta, do_r1=True, grad_clip=cfg.get("grad_clip", 10.0))
        gs = trainers.generator_step(G, D, opt_g, z, cond, meta, gt_segments=gt, ema=ema, generator_kwargs=dict(jitter=jitter))
        for name, v in list(d.items()) + [(k, v) for k, v in gs.items() if k != "topk"]:
            assert bool(torch.isfinite(torch.as_tensor(v)).all()), (it, name)
        assert float(d["r1"]) > 0 and float(d["gan"]) > 0 and float(gs["gan"]) > 0
    moved_g = [n for n, p in G.named_parameters() if p.grad is not None and not torch.equal(p.detach(), g0[n])]
    with_grad_g = [n for n, p in G.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0]
    assert len(with_grad_g) > 200 and set(with_grad_g) <= set(moved_g), sorted(set(with_grad_g) - set(moved_g))[:5]
    moved_d = [n for n, p in D.named_parameters() if p.grad is not None and not torch.equal(p.detach(), d0[n])]
    with_grad_d = [n for n, p in D.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0]
    assert len(with_grad_d) > 50 and set(with_grad_d) <= set(moved_d)
    assert ema.num_updates == 2
    assert torch.cuda.max_memory_allocated() < 100e9
    # the discriminator treats the samples of a batch independently (no batch statistics anywhere in it).  Eval mode (in
    # train mode every call runs a spectral-norm power iteration, i.e. changes the weights), after the iterations above (a
    # fresh module's u / v are random: sigma = u^T W v is then meaningless and the network overflows)
    D.eval()
    with torch.no_grad():
        whole = D(real, None, 1.0)["prediction"]
        one = D(real[2:3], None, 1.0)["prediction"]
    assert bool(torch.isfinite(whole).all()) and rel_err(one.cpu(), whole[2:3].cpu()) < 1e-4


def test_differentiable_path_matches_the_inference_engines_at_the_config4_geometry():
    """The training path (library GEMMs + HIP adjoint kernels) and the fused inference engines are two implementations of one
    function in eval mode; at config 4's geometry they must agree inside the parity budget (the inference engines are the ones
    pinned to the reference's vectors / the oracle at BASELINE sizes by tests/test_gpu_baseline_workloads.py)."""
    G, cfg, z, cond, jitter, _, _ = _cfg4(2)
    G.eval()
    fwd = {k: v for k, v in cfg.items() if isinstance(k, str)}
    with torch.no_grad():
        fast = G(z, cond, jitter=jitter, **fwd)
        slow = G(z, cond, jitter=jitter, differentiable=True, **fwd)
    for k in ("rgbs", "rgbs_render"):
        e = rel_err_channels(fast[k].cpu(), slow[k].cpu())
        print(f"config-4 geometry, eval mode: fused engines vs differentiable path, {k}: {e:.2e}")
        assert e < TOL, (k, e)
