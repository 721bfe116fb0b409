"""Golden fixtures for the training-side rows from the real Python reference.

BUILD CONTAINER ONLY (imports /root/reference through _ref_import.py).  Stores data only:
  disc_tiny.npz         UNetDiscriminator (2 blocks) weights (fp16-exact), images, eval-mode outputs, the reference's
                        logistic GAN loss, balanced segmentation loss and R1 penalty on those outputs
  disc_tiny_amp.npz     the same module and inputs under float16 autocast (the reference's AMP mode): outputs, losses, the
                        weight gradient of the D loss
  ema_tiny.npz          ExponentialMovingAverage: parameters before / after three updates, shadow parameters
  ref_ckpt_tiny_*.pth   the files BaseTrainer.save_model writes (pickled generator module, pickled EMA object, optimizer
                        state dict) for a tiny generator -- to pin the checkpoint loader
  param_order.json      named_parameters() order of the three shipped generator configs (EMA shadow lists are positional)
  gstep_tiny.npz/.json  PhaseTrainer.init_optimizer's five Adam groups for a tiny Map3DGenerator (names per group, learning
                        rates) and PhaseTrainer._train_generator run on stand-in generator / discriminator modules: the z it
                        drew, loss, top-k count and the gradients it left on the generator -- pins the G step's loss algebra
  gen_train_mixed_amp.npz   the same train-mode forward + backward under float16 autocast: outputs and gradients only
  gen_train_*.npz       TRAIN-mode generator forward + backward: weights, conditions, every random tensor, the
                        outputs, the gradient of a fixed random projection of the outputs w.r.t. every parameter and z, and
                        the buffers the train-mode forward overwrote (BatchNorm running statistics, spectral-norm u / v)
Run:  python tests/golden/make_golden_train.py
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import _ref_import  # noqa: E402

_ref_import.install()
for name in ("tensorboardX", "torch.utils.tensorboard"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.SummaryWriter = object
        sys.modules[name] = m

import configs as ref_configs  # noqa: E402
import lib.generators.map3d_generator as ref_gen  # noqa: E402
from lib import implicit_funcitions as ref_impl  # noqa: E402
from lib.components.ema import ExponentialMovingAverage  # noqa: E402
from lib.discriminators.unet_discriminators import UNetDiscriminator  # noqa: E402

from make_golden import condition_weights, save, synthetic, tiny_cfg  # noqa: E402


def disc_fixture():
    torch.manual_seed(11)
    kw = dict(latent_dim=16, gen_height=32, gen_width=16, semantic_dim=0, label_dim=3, discriminator_blocks=2)
    D = UNetDiscriminator(**kw).eval()
    sd = D.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if v.is_floating_point():
                v.copy_(v.half().float())                      # fp16-exact weights: the fixture stores them as fp16
            if k.endswith("bias"):
                v.copy_((0.1 * torch.randn(v.shape)).half().float())
        for k, v in sd.items():                                # exact spectral-norm vectors (eval mode uses them as stored)
            if k.endswith("weight_orig"):
                U, S, Vh = torch.linalg.svd(v.flatten(1), full_matrices=False)
                sd[k.replace("weight_orig", "weight_u")].copy_(U[:, 0])
                sd[k.replace("weight_orig", "weight_v")].copy_(Vh[0])
    D.load_state_dict(sd)
    g = torch.Generator().manual_seed(12)
    real = torch.randn(3, 3, 32, 16, generator=g).clamp(-1, 1)
    fake = torch.randn(3, 3, 32, 16, generator=g).clamp(-1, 1)
    real.requires_grad_(True)
    out_real = D(real, None, 1.0)
    out_fake = D(fake, None, 1.0)
    # the trainer's loss functions, called unbound on a minimal stand-in for `self` (CPU: GradScaler disabled => scale 1)
    import lib.trainers.phase_trainer as pt
    me = types.SimpleNamespace(device="cpu", amp=False, scaler=torch.cuda.amp.GradScaler(enabled=False))
    meta = dict(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, label_dim=3)
    r1 = pt.PhaseTrainer._calculate_r1_regularization(me, real, out_real, {"name": "p"}, meta)
    gt = torch.randint(0, 3, (3, 32, 16), generator=g)
    seg_real, acc_real, prob_real = pt.PhaseTrainer._calculate_segmentation_loss(me, out_real["segments"], gt, meta)
    seg_gen, _, prob_gen = pt.PhaseTrainer._calculate_segmentation_loss(me, out_fake["segments"], torch.zeros_like(gt), meta)
    gan = torch.nn.functional.softplus(out_fake["prediction"]).mean() + torch.nn.functional.softplus(-out_real["prediction"]).mean()
    # gradient of the reference's D loss (gan + 4 * r1 + segmentation) w.r.t. one weight: pins the whole D step's backward
    D.zero_grad()
    loss = gan + 4 * r1 + (seg_real + seg_gen)
    loss.backward()
    gkey = "body_down.0.conv2.1.weight_orig"
    state16 = {k: (v.half() if v.is_floating_point() and not k.endswith(("weight_u", "weight_v")) else v) for k, v in sd.items()}
    save("disc_tiny", state=state16, real=real.detach(), fake=fake, gt_segments=gt,
         out_real={k: v.detach() for k, v in out_real.items()}, out_fake={k: v.detach() for k, v in out_fake.items()},
         loss=dict(r1=r1.detach(), gan=gan.detach(), seg_real=seg_real.detach(), seg_gen=seg_gen.detach(), acc_real=acc_real,
                   prob_real=prob_real.detach(), prob_gen=prob_gen.detach(), total=loss.detach()),
         grad={gkey: dict(D.named_parameters())[gkey].grad})
    json.dump(dict(kwargs=kw, meta=meta, grad_key=gkey), open(os.path.join(HERE, "disc_tiny.json"), "w"))
    # ---- the same module, inputs and losses under float16 autocast (the reference's AMP mode, lib/trainers/base_trainer.py:50-51:
    # torch.cuda.amp.autocast around the network forwards; CPU autocast stands in for it here -- same casting policy, the
    # convolutions run in f16 with f16 outputs): disc_tiny_amp.npz holds ONLY the outputs, losses and the weight gradient;
    # weights and inputs are disc_tiny's.  No random number is drawn below, so disc_tiny.npz is bit-identical with or without it.
    D.zero_grad()
    real_a = real.detach().clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.float16):
        a_real = D(real_a, None, 1.0)
        a_fake = D(fake, None, 1.0)
    assert a_real["prediction"].dtype == torch.float16
    a_real32 = {k: v.float() for k, v in a_real.items()}
    a_fake32 = {k: v.float() for k, v in a_fake.items()}
    r1_a = pt.PhaseTrainer._calculate_r1_regularization(me, real_a, a_real32, {"name": "p"}, meta)
    seg_real_a, acc_a, _ = pt.PhaseTrainer._calculate_segmentation_loss(me, a_real32["segments"], gt, meta)
    seg_gen_a, _, _ = pt.PhaseTrainer._calculate_segmentation_loss(me, a_fake32["segments"], torch.zeros_like(gt), meta)
    gan_a = torch.nn.functional.softplus(a_fake32["prediction"]).mean() + torch.nn.functional.softplus(-a_real32["prediction"]).mean()
    loss_a = gan_a + 4 * r1_a + (seg_real_a + seg_gen_a)
    loss_a.backward()
    save("disc_tiny_amp", out_real={k: v.detach() for k, v in a_real32.items()}, out_fake={k: v.detach() for k, v in a_fake32.items()},
         loss=dict(r1=r1_a.detach(), gan=gan_a.detach(), seg_real=seg_real_a.detach(), seg_gen=seg_gen_a.detach(), total=loss_a.detach()),
         grad={gkey: dict(D.named_parameters())[gkey].grad})


def ema_and_checkpoint_fixture():
    cfg = tiny_cfg()
    torch.manual_seed(21)
    G = ref_gen.Map3DGenerator(**cfg)
    ema = ExponentialMovingAverage(G.parameters(), decay=0.999)
    before = [p.detach().clone() for p in G.parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(22)
    for _ in range(3):
        with torch.no_grad():
            for p in G.parameters():
                p.add_(0.01 * torch.randn(p.shape, generator=g))
        ema.update(G.parameters())
    after = [p.detach().clone() for p in G.parameters() if p.requires_grad]
    names = [n for n, p in G.named_parameters() if p.requires_grad]
    pick = [0, 5, len(names) // 2, len(names) - 1]
    save("ema_tiny", **{f"before/{i}": before[i] for i in pick}, **{f"after/{i}": after[i] for i in pick},
         **{f"shadow/{i}": ema.shadow_params[i] for i in pick}, num_updates=np.asarray(ema.num_updates),
         decay=np.asarray(ema.decay))
    # the trainer's files (BaseTrainer.save_model: torch.save of the module / the EMA object / optimizer state dicts)
    G.neural_field_cls = None
    torch.save(G, os.path.join(HERE, "ref_ckpt_tiny_generator.pth"))
    torch.save(ema, os.path.join(HERE, "ref_ckpt_tiny_ema.pth"))
    opt = torch.optim.Adam(G.parameters(), lr=1e-3, betas=(0.0, 0.9))
    torch.save(opt.state_dict(), os.path.join(HERE, "ref_ckpt_tiny_optimizer_G.pth"))
    torch.save({k: v.clone() for k, v in G.state_dict().items()}, os.path.join(HERE, "ref_ckpt_tiny_generator_state_dict.pth"))
    json.dump(dict(meta={k: v for k, v in cfg.items() if k != "neural_field_cls" and isinstance(v, (int, float, str, bool, list, type(None)))},
                   names=names, pick=pick), open(os.path.join(HERE, "ref_ckpt_tiny.json"), "w"))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("ref_ckpt_tiny"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")


def frontend_fixture():
    """the dataset's SMPL -> conditions conversion (lib/data/datasets.py:117-181, _preprocess_smpl_fix_body)
    and the preprocessor's camera math (lib/data/preprocessor.py:72-97, _forward_fix_body), both called UNBOUND on stand-in
    `self` objects (the classes' constructors need the licensed SMPL file / pytorch3d rasteriser) with a synthetic SMPL
    prediction record."""
    import lib.data.datasets as ds
    import lib.data.preprocessor as pp
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(5)
    V, J = 60, 24
    joints_idx = list(rng.permutation(49)[:J])
    pred = {
        "orig_cam": rng.uniform(0.6, 1.2, (1, 4)).astype(np.float64) * np.array([1, 1, 0.1, 0.1]),
        "joints": rng.randn(1, 49, 3),
        "full_pose": Rotation.random(J, random_state=6).as_matrix()[None],
        "tpose_vertices": rng.randn(1, V, 3) * 0.5,
        "fk_matrices": np.concatenate([np.concatenate([Rotation.random(J, random_state=7).as_matrix(), rng.randn(J, 3, 1) * 0.3], 2),
                                       np.tile(np.array([[[0, 0, 0, 1.0]]]), (J, 1, 1))], 1)[None],
        "lbs_weights": (lambda w: w / w.sum(1, keepdims=True))(rng.rand(V, J) ** 4),
        "betas": rng.randn(1, 10),
    }
    me = types.SimpleNamespace(coodinate_mode="fix_body", joints=joints_idx, smpl_tpose_vertices=rng.randn(V, 3).astype(np.float64),
                               inference=True)
    out = ds.SHHQDataset._preprocess_smpl_fix_body(me, pred)
    # batch of 3 identical records with different camera rotations through the preprocessor
    B = 3
    data = {k: torch.from_numpy(np.asarray(v)).float()[None].repeat(B, *([1] * np.asarray(v).ndim)) for k, v in out.items()}
    data["scales"] = data["scales"].reshape(B)
    h = torch.tensor([0.3, -0.5, 0.0])
    v = torch.tensor([0.1, 0.0, -0.2])
    r = torch.tensor([0.0, 0.05, 0.0])
    pself = types.SimpleNamespace(device="cpu")
    res, R_raster = pp.SHHQPreprocessor._forward_fix_body(pself, dict(data), h, v, r)
    save("frontend", pred={k: np.asarray(v) for k, v in pred.items()}, joints_index=np.asarray(joints_idx),
         smpl_tpose_vertices=me.smpl_tpose_vertices, conditions={k: np.asarray(v) for k, v in out.items()},
         angles=dict(h=h, v=v, r=r), cam2world=res["cam2world_matrices"], R_raster=R_raster)


def param_order_fixture():
    out = {}
    for name in ("MAP3DBN", "MAP3DBN512", "MAP3DBN512L"):
        cfg = {k: v for k, v in getattr(ref_configs, name).items() if isinstance(k, str)}
        cfg.update(dataset_length=4)
        cfg["neural_field_cls"] = ref_impl.COORDCONCATSIREN
        G = ref_gen.Map3DGenerator(**cfg)
        out[name] = [[n, list(p.shape)] for n, p in G.named_parameters() if p.requires_grad]
    json.dump(out, open(os.path.join(HERE, "param_order.json"), "w"))
    print("param_order.json", {k: len(v) for k, v in out.items()})


def generator_train_fixture(name, seed, batch=3, nerf_noise=0.3, use_pool=False, amp=False, **over):
    cfg = tiny_cfg(**over)
    torch.manual_seed(seed)
    G = ref_gen.Map3DGenerator(**cfg)
    G.set_device("cpu")
    condition_weights(G, seed)
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():                                  # u, v off the singular vectors: the power iteration must move them
        for k, v in G.state_dict().items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                v.copy_(torch.nn.functional.normalize(v + 0.3 * torch.randn(v.shape, generator=g), dim=0))
        G.latent_pool.latents.copy_(torch.randn(G.latent_pool.latents.shape, generator=g))
    G.train()
    state0 = {k: v.clone() for k, v in G.state_dict().items()}
    cond = synthetic.make_conditions(batch, n_vertices=128, seed=seed, pose_scale=0.6)
    z = torch.randn(batch, cfg["latent_dim"], generator=g).requires_grad_(True)
    idx = torch.tensor([2, 0, 3][:batch]) if use_pool else None
    run = dict(cfg)
    run["nerf_noise"] = nerf_noise
    R, S = cfg["render_height"] * cfg["render_width"], cfg["num_steps"]
    rs = seed + 7
    torch.manual_seed(rs)                                  # the reference's consumption order (make_golden.generator_fixture)
    jitter = torch.rand(batch, R, S, 1)
    torch.randn(batch, 1), torch.randn(batch, 1)
    noise = torch.randn(batch, R, S, 1) * nerf_noise
    torch.manual_seed(rs)
    out = G.forward(z, cond, latent_indices=idx, **run)
    p_rgb = torch.randn(out["rgbs"].shape, generator=g)
    p_render = torch.randn(out["rgbs_render"].shape, generator=g)
    loss = (out["rgbs"] * p_rgb).sum() + (out["rgbs_render"] * p_render).sum()
    loss.backward()
    grads = {n: p.grad for n, p in G.named_parameters() if p.grad is not None}
    if z.grad is not None:
        grads["__z__"] = z.grad
    state1 = G.state_dict()
    changed = {k: v for k, v in state1.items() if not torch.equal(v, state0[k])}
    meta = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool))}
    meta["mod_blocks"] = list(cfg["mod_blocks"])
    meta["nerf_noise"] = nerf_noise
    extra = dict(latent_indices=idx) if use_pool else {}
    save(name, state=state0, cond=cond, z=z, jitter=jitter, noise=noise, p_rgb=p_rgb, p_render=p_render,
         meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
         out=dict(rgbs=out["rgbs"], rgbs_render=out["rgbs_render"], loss=loss), grad=grads, buffers_after=changed, **extra)
    print("   parameters with gradient:", len(grads), "of", len(list(G.named_parameters())), "| buffers changed:", len(changed))
    if not amp:
        return
    # ---- the same module, inputs and random tensors under float16 autocast (the reference's AMP mode; CPU autocast stands in for
    # torch.cuda.amp.autocast: same casting policy).  <name>_amp.npz holds only outputs and gradients; everything else is <name>'s.
    G.load_state_dict(state0)
    G.zero_grad()
    z2 = z.detach().clone().requires_grad_(True)
    torch.manual_seed(rs)
    with torch.autocast("cpu", dtype=torch.float16):
        out2 = G.forward(z2, cond, latent_indices=idx, **run)
        loss2 = (out2["rgbs"].float() * p_rgb).sum() + (out2["rgbs_render"].float() * p_render).sum()
    loss2.backward()
    grads2 = {n: p.grad.float() for n, p in G.named_parameters() if p.grad is not None}
    save(name + "_amp", out=dict(rgbs=out2["rgbs"].float(), rgbs_render=out2["rgbs_render"].float(), loss=loss2), grad=grads2)


def gstep_fixture():
    import lib.trainers.phase_trainer as pt
    # ---- optimiser groups of a real (tiny) generator
    cfg = tiny_cfg()
    torch.manual_seed(41)
    G = ref_gen.Map3DGenerator(**cfg)
    kw = dict(latent_dim=16, gen_height=32, gen_width=16, semantic_dim=0, label_dim=3, discriminator_blocks=2)
    D = UNetDiscriminator(**kw).eval()
    sd = D.state_dict()
    with torch.no_grad():                                      # as in disc_fixture: fp16-exact weights, exact spectral-norm vectors
        for k, v in sd.items():
            if v.is_floating_point():
                v.copy_(v.half().float())
        for k, v in sd.items():
            if k.endswith("weight_orig"):
                U, S, Vh = torch.linalg.svd(v.flatten(1), full_matrices=False)
                sd[k.replace("weight_orig", "weight_u")].copy_(U[:, 0])
                sd[k.replace("weight_orig", "weight_v")].copy_(Vh[0])
    D.load_state_dict(sd)
    meta = dict(gen_lr=5e-5, disc_lr=2e-4, betas=(0.0, 0.9), weight_decay=0, appearance_codes_lr_mul=1.0,
                mapping_net_lr_mul=0.05, neural_field_lr_mul=0.05)
    me = types.SimpleNamespace(generator_ddp=G, discriminator_ddp=D, output_dir="/nonexistent", device="cpu")
    pt.PhaseTrainer.init_optimizer(me, meta)
    names = {id(p): n for n, p in G.named_parameters()}
    groups = [dict(name=g["name"], lr=g["lr"], params=[names[id(p)] for p in g["params"]]) for g in me.optimizer_G.param_groups]

    # ---- _train_generator on stand-in networks
    class StubG(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(16, 3 * 32 * 16)
            self.latent_pool = torch.nn.Embedding(4, 16)

        def forward(self, z, conditions, disable_synthesis=False, latent_indices=None, **kw):
            img = torch.tanh(self.lin(z)).view(z.shape[0], 3, 32, 16)
            return {"rgbs": img, "rgbs_render": img[:, :, ::4, ::4]}

    torch.manual_seed(42)
    sg = StubG()
    B = 5
    g = torch.Generator().manual_seed(43)
    data = dict(images=torch.randn(B, 3, 32, 16, generator=g), latents=torch.randn(B, 16, generator=g),
                rasterized_segments=torch.randint(0, 3, (B, 32, 16), generator=g), indices=torch.arange(B) % 4)
    tmeta = dict(gan_lambda=1.0, segmentation_lambda=1.0, latent_dim=16, z_dist="gaussian", latent_lambda=0,
                 perceptual_lambda=[0, 0, 0, 0], photometric_lambda=0, label_dim=3, topk_interval=2000, topk_v=0.5)
    phase = dict(uncond=True, gen_modal="rgbs", rotate=True, name="p")
    me2 = types.SimpleNamespace(device="cpu", amp=False, batch_split=1, scaler=torch.cuda.amp.GradScaler(enabled=False),
                                generator_ddp=sg, discriminator_ddp=D, discriminator=types.SimpleNamespace(step=40000))
    me2._get_disc_input_gen = types.MethodType(pt.PhaseTrainer._get_disc_input_gen, me2)
    me2._calculate_segmentation_loss = types.MethodType(pt.PhaseTrainer._calculate_segmentation_loss, me2)
    torch.manual_seed(44)
    z = torch.randn((B, 16))                                  # what z_sampler will draw
    torch.manual_seed(44)
    loss, topk = pt.PhaseTrainer._train_generator(me2, data, 1.0, tmeta, phase)
    disc16 = {k: (v.half() if v.is_floating_point() and not k.endswith(("weight_u", "weight_v")) else v)
              for k, v in D.state_dict().items()}
    save("gstep_tiny", stub={k: v for k, v in sg.state_dict().items()}, disc=disc16, data=data, z=z,
         loss=torch.tensor(loss), topk=torch.tensor(topk), grad={n: p.grad for n, p in sg.named_parameters() if p.grad is not None})
    json.dump(dict(groups=groups, disc_kwargs=kw, meta=tmeta, d_step=40000, generator_meta={k: v for k, v in cfg.items()
                                                                                         if isinstance(v, (int, float, str, bool))}
                   | {"mod_blocks": list(cfg["mod_blocks"])}),
              open(os.path.join(HERE, "gstep_tiny.json"), "w"), indent=0)
    print("   groups:", [(g["name"], len(g["params"]), g["lr"]) for g in groups], "| loss", loss, "topk", topk)


if __name__ == "__main__":
    if "--only-gstep" in sys.argv:
        gstep_fixture()
        sys.exit(0)
    if "--only-gen-train" in sys.argv:
        generator_train_fixture("gen_train_mixed", 31)
        generator_train_fixture("gen_train_isolated_legacy_pool", 37, use_pool=True, legacy_mode=True, map3d_mode="isolated",
                                white_back=True, last_back=True, clamp_mode="softplus")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "disc":
        disc_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gen_amp":
        generator_train_fixture("gen_train_mixed", 31, amp=True)
        sys.exit(0)
    generator_train_fixture("gen_train_mixed", 31, amp=True)
    generator_train_fixture("gen_train_isolated_legacy_pool", 37, use_pool=True, legacy_mode=True, map3d_mode="isolated",
                            white_back=True, last_back=True, clamp_mode="softplus")
    gstep_fixture()
    frontend_fixture()
    disc_fixture()
    ema_and_checkpoint_fixture()
    param_order_fixture()
