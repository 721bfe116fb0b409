"""World-size-2 (and 3) gloo tests of the batch-sharding helpers used by bench.py / the app (CPU only)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

par = importlib.import_module("3dhumangan_amd.parallel")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4)
        cond = {"a": torch.arange(total * 6, dtype=torch.float32).reshape(total, 2, 3), "s": torch.arange(total).float()}
        zl, cl = par.shard_batch(z, cond)
        lo, hi = par.shard_bounds(total, rank, world)
        assert torch.equal(zl, z[lo:hi]) and torch.equal(cl["a"], cond["a"][lo:hi]) and torch.equal(cl["s"], cond["s"][lo:hi])
        # "generate": an image that encodes the global sample index
        imgs = zl[:, :1, None, None].expand(-1, 3, 2, 2).contiguous() + 0.5
        full = par.gather_images(imgs, total)
        want = z[:, :1, None, None].expand(-1, 3, 2, 2) + 0.5
        assert torch.equal(full, want)
        slow = par.max_over_ranks(1.0 + rank)
        assert slow == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 5), (3, 7)])
def test_shard_and_gather(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            spans = [par.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------- R1 all-gather + discriminator step, world size 2

def _r1_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
        trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
        torch.manual_seed(3)                                   # same weights and same global batch on every rank
        D = disc.UNetDiscriminator(latent_dim=8, gen_height=16, gen_width=8, label_dim=2, discriminator_blocks=2).eval()
        g = torch.Generator().manual_seed(4)
        total = 5                                              # uneven shards: 3 + 2
        real, fake = torch.randn(total, 3, 16, 8, generator=g), torch.randn(total, 3, 16, 8, generator=g)
        gt = torch.zeros(total, 16, 8, dtype=torch.long)
        meta = dict(gan_lambda=1.0, segmentation_lambda=0.0, r1_lambda=10.0, label_dim=2)
        # single-process truth on the whole batch
        ref = disc.UNetDiscriminator(latent_dim=8, gen_height=16, gen_width=8, label_dim=2, discriminator_blocks=2).eval()
        ref.load_state_dict(D.state_dict())
        r_ref = trainers.discriminator_step(ref, torch.optim.SGD(ref.parameters(), lr=0.0), real, fake, gt, meta,
                                            r1_mode="per_sample")
        # sharded step: gathered statistics identical on both ranks, gradients equal to the whole-batch ones
        lo, hi = par.shard_bounds(total, rank, world)
        stat = torch.arange(lo, hi, dtype=torch.float32, requires_grad=True) * 1.0
        gathered = par.r1_allgather(stat)
        assert torch.equal(gathered.detach(), torch.arange(total, dtype=torch.float32))
        gathered.mean().backward()                              # only the local slice carries a graph
        # R1 only (gan / segmentation are per-shard means: with uneven shards their rank average is not the whole-batch mean)
        meta_r1 = dict(meta, gan_lambda=1.0)
        r_loc = trainers.discriminator_step(D, torch.optim.SGD(D.parameters(), lr=0.0), real[lo:hi], fake[lo:hi], gt[lo:hi], meta_r1,
                                            distributed=True, r1_mode="per_sample")
        assert abs(float(r_loc["r1"]) - float(r_ref["r1"])) <= 1e-5 * abs(float(r_ref["r1"])), (r_loc["r1"], r_ref["r1"])
        # even shards (the first 4 samples): every discriminator gradient equals the single-process whole-batch gradient
        ref.zero_grad(set_to_none=True)
        trainers.discriminator_step(ref, torch.optim.SGD(ref.parameters(), lr=0.0), real[:4], fake[:4], gt[:4], meta,
                                    r1_mode="per_sample")
        lo, hi = par.shard_bounds(4, rank, world)
        trainers.discriminator_step(D, torch.optim.SGD(D.parameters(), lr=0.0), real[lo:hi], fake[lo:hi], gt[lo:hi], meta,
                                    distributed=True, r1_mode="per_sample")
        worst = 0.0
        checked = 0
        for (n, a), (_, b) in zip(D.named_parameters(), ref.named_parameters()):
            assert (a.grad is None) == (b.grad is None), n
            if b.grad is not None:                              # heads that do not feed this loss have no gradient
                worst = max(worst, float((a.grad - b.grad).abs().max() / b.grad.abs().max().clamp_min(1e-12)))
                checked += 1
        assert checked > 20 and worst < 1e-4, (checked, worst)
        # the default ("reference") statistic, sharded: the penalty is the rank mean of the per-rank reference penalties and
        # the gradients the rank mean of the per-rank gradients -- what DDP's averaging gives the reference
        per_rank = []
        for r in range(world):
            a0, a1 = par.shard_bounds(4, r, world)
            ref.zero_grad(set_to_none=True)
            rr = trainers.discriminator_step(ref, torch.optim.SGD(ref.parameters(), lr=0.0), real[a0:a1], fake[a0:a1], gt[a0:a1], meta)
            per_rank.append((float(rr["r1"]), {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}))
        r_def = trainers.discriminator_step(D, torch.optim.SGD(D.parameters(), lr=0.0), real[lo:hi], fake[lo:hi], gt[lo:hi], meta,
                                            distributed=True)
        want = sum(v for v, _ in per_rank) / world
        assert abs(float(r_def["r1"]) - want) <= 1e-5 * abs(want), (r_def["r1"], want)
        worst = 0.0
        for n, p in D.named_parameters():
            if p.grad is not None:
                w = sum(gr[n] for _, gr in per_rank) / world
                worst = max(worst, float((p.grad - w).abs().max() / w.abs().max().clamp_min(1e-12)))
        assert worst < 1e-4, worst
        # rank-variant gradient sets: a parameter with a gradient on rank 0 only still travels (zeros from the others), one
        # without a gradient anywhere stays None
        a, b, c = (torch.nn.Parameter(torch.ones(3)) for _ in range(3))
        a.grad = torch.full((3,), float(rank + 1))
        if rank == 0:
            b.grad = torch.full((3,), 4.0)
        par.allreduce_gradients([a, b, c], average=True)
        assert torch.equal(a.grad, torch.full((3,), 1.5)) and torch.equal(b.grad, torch.full((3,), 2.0)) and c.grad is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_r1_allgather_and_sharded_discriminator_step_world2():
    """north_star: batch-dim sharding with an all-gather for the discriminator R1 step.  Two ranks, half the batch each: the
    gathered per-sample statistics are the global ones on both ranks, the penalty equals the whole-batch penalty, and after
    the gradient all-reduce every discriminator gradient equals the single-process whole-batch gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_r1_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


# ---------------------------------------------------------------- generator step: synchronised BatchNorm, world size 2 (gloo)

def _syncbn_worker(rank, world, port, q):
    """The synthesis half of the differentiable generator (pure torch + collectives, so it runs on CPU) on a batch shard:
    with the BatchNorm moments all-reduced, outputs, averaged gradients and the updated running statistics equal the
    single-process whole-batch step -- which is itself pinned to the reference's autograd by the golden train fixtures."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        from conftest import load_golden
        from _torch_spade_kernels import TorchKernels
        tk = TorchKernels()                          # stand-in arithmetic: the test is about the collective algebra
        gens = importlib.import_module("3dhumangan_amd.lib.generators")
        impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
        diff = importlib.import_module("3dhumangan_amd.lib.generators.differentiable")
        g = load_golden("gen_train_mixed")
        cfg = dict(g["meta"])
        cfg["neural_field_cls"] = impl.COORDCONCATSIREN

        def make():
            G = gens.Map3DGenerator(**cfg)
            G.load_state_dict(g["state"], strict=True)
            return G.train()

        B, Fd = 4, cfg["feature_dim"]
        rhw, ghw = (cfg["render_height"], cfg["render_width"]), (cfg["gen_height"], cfg["gen_width"])
        gen = torch.Generator().manual_seed(5)
        fmap = torch.randn(B, rhw[0] * rhw[1], Fd, generator=gen)
        styles = torch.randn(B, 1, Fd, generator=gen)
        proj = torch.randn(B, 3, *ghw, generator=gen)
        whole = make()
        f_all = fmap.clone().requires_grad_(True)
        # single-process reference: the whole batch with the collective path disabled (group=False)
        out_all = diff.synthesis_forward(whole, f_all, styles, rhw, ghw, training=True, group=False, spade_kernels=tk)
        (out_all * proj).sum().backward()
        lo, hi = par.shard_bounds(B, rank, world)
        mine = make()
        f_loc = fmap[lo:hi].clone().requires_grad_(True)
        out = diff.synthesis_forward(mine, f_loc, styles[lo:hi], rhw, ghw, training=True, group=dist.group.WORLD,
                                     spade_kernels=tk)
        (out * proj[lo:hi]).sum().backward()
        par.allreduce_gradients(list(mine.parameters()), average=False)
        assert float((out - out_all[lo:hi]).abs().max() / out_all.abs().max()) < 1e-5
        assert float((f_loc.grad - f_all.grad[lo:hi]).abs().max() / f_all.grad.abs().max()) < 1e-4
        worst, checked = 0.0, 0
        for (n, a), (_, b) in zip(mine.named_parameters(), whole.named_parameters()):
            if b.grad is None or float(b.grad.abs().max()) < 1e-4:
                continue
            worst = max(worst, float((a.grad - b.grad).abs().max() / b.grad.abs().max()))
            checked += 1
        assert checked > 100 and worst < 2e-4, (checked, worst)
        sa, sb = mine.state_dict(), whole.state_dict()
        for k in sb:
            if "running_" in k or k.endswith(("weight_u", "weight_v")):
                assert float((sa[k] - sb[k]).abs().max()) < 1e-5 * (1 + float(sb[k].abs().max())), k
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_generator_synthesis_step_with_synchronised_batchnorm_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


# ---------------------------------------------------------------- bench.py's distributed plumbing, world size 2 (gloo)

def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import sys
        import time
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        r, w, local, dist_on = bench.dist_env()
        assert (r, w, local, dist_on) == (rank, world, rank, True)
        bench.init_distributed(local, backend="gloo")
        calls = []

        def step():
            calls.append(1)
            time.sleep(0.01 * (1 + rank))                      # rank 1 is the slow one

        dt = bench.timed_loop(step, steps=5, warmup=2, dist_on=True, device="cpu")
        assert len(calls) == 7                                  # warm-up steps are run, but only `steps` are timed
        assert 0.09 < dt < 1.0, dt                              # the MAX over ranks: >= 5 * 0.02 s, the slow rank's time
        dts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(dts, torch.tensor([dt], dtype=torch.float64))
        assert all(float(t) == dt for t in dts)                 # every rank reports the same (max) time
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_bench_distributed_plumbing_world2():
    """bench.py's N > 1 path (rendezvous from the torchrun environment, barrier-bracketed timed loop, MAX over ranks) with the
    gloo backend: the first multi-GPU run of the driver must not fail on plumbing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


# ---------------------------------------------------------------- one whole adversarial iteration (bench.py --mode trainstep), world size 2

class _SynthesisOnlyGenerator(torch.nn.Module):
    """Stand-in for the differentiable generator on CPU: a real Map3DGenerator whose forward skips the (HIP-only) field and
    feeds its own SPADE synthesis network -- lib/generators/differentiable.synthesis_forward with the plain-torch kernel set,
    batch-statistics BatchNorm synchronised over `group`, spectral-norm power iterations -- from a linear map of z.  Same
    parameter set, names and optimiser groups as the product's generator."""

    def __init__(self, cfg, state, group):
        super().__init__()
        gens = importlib.import_module("3dhumangan_amd.lib.generators")
        self.inner = gens.Map3DGenerator(**cfg)
        self.inner.load_state_dict(state, strict=True)
        self.inner.train()
        self.group = group
        self.rhw = (cfg["render_height"], cfg["render_width"])
        self.ghw = (cfg["gen_height"], cfg["gen_width"])
        g = torch.Generator().manual_seed(17)
        self.fmap_proj = torch.nn.Parameter(torch.randn(cfg["latent_dim"], self.rhw[0] * self.rhw[1] * cfg["feature_dim"], generator=g) * 0.2)
        self.style_proj = torch.nn.Parameter(torch.randn(cfg["latent_dim"], cfg["feature_dim"], generator=g) * 0.5)

    def forward(self, z, conditions, **kw):
        from _torch_spade_kernels import TorchKernels
        diff = importlib.import_module("3dhumangan_amd.lib.generators.differentiable")
        B = z.shape[0]
        fmap = (z @ self.fmap_proj).reshape(B, self.rhw[0] * self.rhw[1], -1)
        styles = (z @ self.style_proj).reshape(B, 1, -1)
        rgb = diff.synthesis_forward(self.inner, fmap, styles, self.rhw, self.ghw, training=True, group=self.group,
                                     spade_kernels=TorchKernels())
        return {"rgbs": rgb}


def _iteration_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from conftest import load_golden
        r, w, local, dist_on = bench.dist_env()
        bench.init_distributed(local, backend="gloo")
        impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
        disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
        trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
        ema_mod = importlib.import_module("3dhumangan_amd.lib.components.ema")
        g = load_golden("gen_train_mixed")
        cfg = dict(g["meta"])
        cfg["neural_field_cls"] = impl.COORDCONCATSIREN
        H, W = cfg["gen_height"], cfg["gen_width"]
        meta = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
        meta.update(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, gen_lr=5e-3, betas=(0.0, 0.9), label_dim=2,
                    equal_shards=True, grad_clip=10.0)
        total = 4
        gen = torch.Generator().manual_seed(11)
        z = torch.randn(total, cfg["latent_dim"], generator=gen)
        real = torch.randn(total, 3, H, W, generator=gen).clamp(-1, 1)
        gt = torch.randint(0, 2, (total, H, W), generator=gen)

        def run(group, lo, hi, distributed):
            G = _SynthesisOnlyGenerator(cfg, g["state"], group)
            torch.manual_seed(99)
            D = disc.UNetDiscriminator(latent_dim=8, gen_height=H, gen_width=W, label_dim=2, discriminator_blocks=2)
            opt_d = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.9))
            opt_g = trainers.make_generator_optimizer(G, meta)
            ema = ema_mod.ExponentialMovingAverage(G.parameters(), decay=0.5)
            out = {}

            def step():
                out["d"], out["g"] = trainers.adversarial_iteration(
                    G, D, opt_d, opt_g, z[lo:hi], {}, real[lo:hi], gt[lo:hi], meta, ema=ema, distributed=distributed,
                    grad_clip=10.0, r1_mode="per_sample")

            # through bench.py's own loop: barrier-bracketed, MAX over ranks; two iterations (the second one runs on the
            # first one's updated weights, running statistics and Adam state)
            dt = bench.timed_loop(step, steps=2, warmup=0, dist_on=True, device="cpu")
            assert dt > 0
            return G, D, ema, out

        G1, D1, ema1, o1 = run(False, 0, total, False)              # single process, whole batch, no collective
        lo, hi = par.shard_bounds(total, rank, world)
        G2, D2, ema2, o2 = run(dist.group.WORLD, lo, hi, True)      # this rank's shard
        assert G2.inner is not None and par.reducer_of(D2).active and par.reducer_of(G2).active
        assert len(par.reducer_of(D2).buckets) >= 1

        def worst(a_mod, b_mod):
            """Largest relative difference over the floating-point state.  Parameters whose gradient is mathematically zero (a
            conv bias in front of a batch-statistics BatchNorm) are left out: their gradient is rounding noise, and Adam turns
            noise of either sign into a full +-lr step."""
            w, n = 0.0, 0
            sa, sb = a_mod.state_dict(), b_mod.state_dict()
            grads = {k: p.grad for k, p in b_mod.named_parameters()}
            for k in sb:
                if not sb[k].dtype.is_floating_point:
                    continue
                gk = grads.get(k)
                if k in grads and (gk is None or float(gk.abs().max()) < 1e-5):
                    continue
                w = max(w, float((sa[k] - sb[k]).abs().max() / (1e-6 + sb[k].abs().max())))
                n += 1
            return w, n

        wd, nd = worst(D2, D1)          # discriminator weights after two Adam steps on all-reduced gradients
        wg, ng = worst(G2, G1)          # generator weights, BatchNorm running statistics, spectral-norm vectors
        assert nd > 20 and wd < 2e-3, (nd, wd)
        assert ng > 100 and wg < 2e-3, (ng, wg)
        live = [p for p in G1.parameters() if p.requires_grad]
        assert len(live) == len(ema1.shadow_params)
        we = max(float((a - b).abs().max() / (1e-6 + b.abs().max())) for a, b, p in zip(ema2.shadow_params, ema1.shadow_params, live)
                 if p.grad is not None and float(p.grad.abs().max()) >= 1e-5)
        assert we < 2e-3, we
        # the R1 penalty is the global-batch one on every rank; the per-shard loss terms average to the whole-batch ones
        assert abs(float(o2["d"]["r1"]) - float(o1["d"]["r1"])) < 1e-3 * (1e-6 + abs(float(o1["d"]["r1"])))
        for part in ("gan", "segmentation"):
            for step_name in ("d", "g"):
                mine = torch.tensor([float(o2[step_name][part])], dtype=torch.float64)
                dist.all_reduce(mine)
                assert abs(float(mine) / world - float(o1[step_name][part])) < 1e-3 * (1e-6 + abs(float(o1[step_name][part]))), (step_name, part)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_whole_adversarial_iteration_world2_matches_the_single_process_whole_batch():
    """one whole `bench.py --mode trainstep` iteration pair -- generator forward with synchronised BatchNorm,
    discriminator step with the R1 all-gather (equal-shard fast path) and the gradient all-reduce overlapped with backward
    (parallel.GradReducer), generator step with its own overlapped all-reduce, Adam on both, EMA -- on two gloo ranks holding
    half the batch each, through bench.py's timed loop, against the same two iterations on the whole batch in one process:
    weights, running statistics, spectral-norm vectors, EMA shadow and losses agree."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_iteration_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


# ---------------------------------------------------------------------------------------------------------------------
# GradReducer on its own (round 5): 25 MB-style bucketing, one multi-tensor copy per bucket, the arrival-order rebuild with
# the never-used parameters last, hooks removed on replacement, and the documented error for a second backward.


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(3)

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.unused_head = torch.nn.Linear(8, 8)           # registered FIRST -> reversed order puts it in the LAST bucket
                self.a = torch.nn.Linear(16, 32)
                self.b = torch.nn.Linear(32, 32)
                self.c = torch.nn.Linear(32, 4)
                self.unused_tail = torch.nn.Linear(4, 4)           # registered LAST -> bucket 0 of the first step: never ready

            def forward(self, x):
                return self.c(torch.relu(self.b(torch.relu(self.a(x)))))

        net = Net()
        ref = Net()
        ref.load_state_dict(net.state_dict())
        x_all = torch.randn(8, 16, generator=torch.Generator().manual_seed(5))
        lo, hi = par.shard_bounds(8, rank, world)
        red = par.GradReducer(net.parameters(), bucket_bytes=2048)          # tiny buckets: several of them
        assert red.active and len(red.buckets) >= 3
        first_cut = [[id(p) for p in b["params"]] for b in red.buckets]
        assert id(net.unused_tail.bias) in first_cut[0]                     # the never-ready parameter leads the first step
        launches = []
        orig = red._launch
        red._launch = lambda b: (launches.append(([id(x) for x in red.buckets].index(id(b)), red._armed)), orig(b))[1]

        def step(model, xs, reducer=None):
            model.zero_grad(set_to_none=True)
            if reducer:
                reducer.prepare()
            (model(xs) ** 2).mean().backward()
            if reducer:
                reducer.finish()

        step(ref, x_all)                                                    # whole batch, one process
        step(net, x_all[lo:hi], red)                                        # this rank's shard
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if "unused" in n:
                assert p.grad is None, n                                    # no gradient on any rank: stays None, as under DDP
            else:
                assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7), n
        assert all(not armed for _, armed in launches), "first step: bucket 0 never completes, everything waits for finish()"
        # rebuilt: arrival order, the unused parameters in a tail of their own
        cut = [[id(p) for p in b["params"]] for b in red.buckets]
        unused = {id(p) for n, p in net.named_parameters() if "unused" in n}
        assert cut != first_cut and set(cut[-1]) <= unused and not (set(sum(cut[:-1], [])) & unused)
        assert id(net.c.bias) in cut[0] or id(net.c.weight) in cut[0]       # the last layer's gradients arrive first
        order = torch.tensor([red._index[i] for b in cut for i in b])
        ranks = [torch.empty_like(order) for _ in range(world)]
        dist.all_gather(ranks, order)
        assert all(torch.equal(ranks[0], t) for t in ranks), "every rank cuts the same buckets"
        # second step: buckets are reduced from inside backward now
        launches.clear()
        step(ref, x_all)
        step(net, x_all[lo:hi], red)
        assert sum(1 for _, armed in launches if armed) >= len(red.buckets) - 1, launches
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if "unused" not in n:
                assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7), n
        # two backward passes between prepare() and finish(): the second one's gradients would miss the reduction -> error
        net.zero_grad(set_to_none=True)
        red.prepare()
        (net(x_all[lo:hi]) ** 2).mean().backward()
        try:
            (net(x_all[lo:hi]) ** 2).mean().backward()
            raise AssertionError("a gradient arriving after its bucket was reduced must raise")
        except RuntimeError as e:
            assert "after its bucket was reduced" in str(e)
        red.finish()                                                        # the collectives already launched are matched on every rank
        # reducer_of: cached per module, replaced (old hooks removed) when the requires_grad set changes
        r1 = par.reducer_of(net, bucket_bytes=2048)
        assert par.reducer_of(net) is r1
        for p in net.a.parameters():
            p.requires_grad_(False)
        r2 = par.reducer_of(net, bucket_bytes=2048)
        assert r2 is not r1 and not r1.active and not r1._hooks
        assert all(id(p) not in r2._slot for p in net.a.parameters())
        step(net, x_all[lo:hi], r2)
        assert net.a.weight.grad is None and net.b.weight.grad is not None
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_grad_reducer_buckets_rebuild_and_double_backward_error_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in got:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_ddp_bucket_size_is_the_reference_default():
    assert par.DDP_BUCKET_BYTES == 25 << 20
    assert par.GradReducer.__init__.__defaults__[2] == par.DDP_BUCKET_BYTES
