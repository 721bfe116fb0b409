"""CPU tests of the generator step's host logic against PhaseTrainer itself (tests/golden/gstep_tiny.*): the five Adam groups of
init_optimizer, and the loss algebra of _train_generator (GAN term over the top-k fakes, balanced segmentation term, backward)
run on stand-in generator / discriminator modules."""
import importlib
import json
import os

import torch

from conftest import GOLDEN, load_golden, rel_err

trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
disc = importlib.import_module("3dhumangan_amd.lib.discriminators")


def _info():
    return json.load(open(os.path.join(GOLDEN, "gstep_tiny.json")))


def test_optimizer_groups_match_the_reference_trainer():
    info = _info()
    cfg = dict(info["generator_meta"])
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    G = gens.Map3DGenerator(**cfg)
    meta = dict(gen_lr=5e-5, betas=(0.0, 0.9), weight_decay=0, appearance_codes_lr_mul=1.0, mapping_net_lr_mul=0.05,
                neural_field_lr_mul=0.05)
    names = {id(p): n for n, p in G.named_parameters()}
    groups = trainers.generator_param_groups(G, meta)
    assert [g["name"] for g in groups] == [g["name"] for g in info["groups"]]
    for mine, ref in zip(groups, info["groups"]):
        assert [names[id(p)] for p in mine["params"]] == ref["params"], mine["name"]       # same members, same ORDER
        assert abs(mine["lr"] - ref["lr"]) < 1e-12, mine["name"]
    opt = trainers.make_generator_optimizer(G, meta)
    assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in info["groups"]]
    assert opt.defaults["betas"] == (0.0, 0.9)


class _StubG(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(16, 3 * 32 * 16)
        self.latent_pool = torch.nn.Embedding(4, 16)

    def forward(self, z, conditions, disable_synthesis=False, latent_indices=None, **kw):
        img = torch.tanh(self.lin(z)).view(z.shape[0], 3, 32, 16)
        return {"rgbs": img, "rgbs_render": img[:, :, ::4, ::4]}


def test_generator_step_loss_and_gradients_match_train_generator():
    info, g = _info(), load_golden("gstep_tiny")
    G = _StubG()
    G.load_state_dict(g["stub"])
    D = disc.UNetDiscriminator(**info["disc_kwargs"]).eval()
    D.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["disc"].items()})
    opt = torch.optim.SGD(G.parameters(), lr=0.0)
    res = trainers.generator_step(G, D, opt, g["z"], {}, dict(info["meta"]), gt_segments=g["data"]["rasterized_segments"],
                                  d_step_count=info["d_step"])
    assert res["topk"] == int(g["topk"])
    assert abs(float(res["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for n, p in G.named_parameters():
        if n in g["grad"]:
            assert rel_err(p.grad, g["grad"][n]) < 1e-4, n
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n


def test_discriminator_weight_gradients_are_skipped_unless_asked_for():
    """The reference computes D's weight gradients in the generator step and zeroes them unread (phase_trainer.py:302, 326-337); the
    default here does not compute them: same loss, same generator gradients, D's parameters trainable again afterwards."""
    info, g = _info(), load_golden("gstep_tiny")
    runs = {}
    for with_d in (False, True):
        G = _StubG()
        G.load_state_dict(g["stub"])
        D = disc.UNetDiscriminator(**info["disc_kwargs"]).eval()
        D.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["disc"].items()})
        res = trainers.generator_step(G, D, torch.optim.SGD(G.parameters(), lr=0.0), g["z"], {}, dict(info["meta"]),
                                      gt_segments=g["data"]["rasterized_segments"], d_step_count=info["d_step"], d_param_grads=with_d)
        assert all(p.requires_grad for p in D.parameters())
        got = [p.grad is not None for p in D.parameters()]
        assert any(got) if with_d else not any(got)
        runs[with_d] = (float(res["loss"]), {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None})
    assert runs[False][0] == runs[True][0]
    assert runs[False][1].keys() == runs[True][1].keys()
    for n, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][n]), n


def test_topk_schedule():
    meta = dict(topk_interval=2000, topk_v=0.6)
    assert trainers.g_step.topk_count({}, 123, 8) == 8                       # no schedule in the config: the whole batch
    assert trainers.g_step.topk_count(meta, 0, 8) == 8
    assert trainers.g_step.topk_count(meta, 40000, 8) == 7                  # 0.99 ** 20 = 0.818
    assert trainers.g_step.topk_count(meta, 400000, 8) == 5                 # floor at topk_v
