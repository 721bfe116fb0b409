#!/usr/bin/env python
"""bench.py -- generator images/sec at 512^2 on N MI355X (one process per GPU) + kernel rooflines.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one Map3DGenerator.forward over a batch of synthetic inputs (random-init weights of the MAP3DBN512
architecture, procedural SMPL-like pose, N(0,1) latents, U(0,1) jitter), everything resident in HBM.  The workload
is BASELINE.json config 3: 512x512 output, 96x96 rays, 64 samples per ray, batch 16 per GPU, hidden width 256 (the
reference's native 512x256 aspect is reported next to it as extra.native_512x256).  Inference shards over the batch
with no data-path collective: weak scaling, value = all images of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TF = 2500.0      # dense f16 / bf16 matrix peak (v_mfma_f32_32x32x16_{f16,bf16})


def build_generator(cfg_name, gen_hw, render_hw, steps, device):
    configs = importlib.import_module("3dhumangan_amd.configs")
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
    cfg = {k: v for k, v in getattr(configs, cfg_name).items() if isinstance(k, str)}
    cfg.update(gen_height=gen_hw[0], gen_width=gen_hw[1], render_height=render_hw[0], render_width=render_hw[1],
               num_steps=steps, dataset_length=4, nerf_noise=0, last_back=cfg["eval_last_back"])
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(1234)
    G = gens.Map3DGenerator(**cfg).to(device).eval()
    G.set_device(device)
    return G, cfg


def make_inputs(cfg, batch, device, seed=1234):
    synthetic = importlib.import_module("3dhumangan_amd.synthetic")
    g = torch.Generator().manual_seed(seed)
    cond = {k: v.to(device) for k, v in synthetic.make_conditions(batch, 6890, seed=seed % 1000).items()}
    z = torch.randn(batch, cfg["latent_dim"], generator=g).to(device)
    R = cfg["render_height"] * cfg["render_width"]
    jitter = torch.rand(batch, R, cfg["num_steps"], 1, generator=g).to(device)
    return z, cond, jitter


def dist_env():
    """(rank, world, local_rank, dist_on) from the torchrun environment; under torchrun always go through the collective
    backend, even with one rank."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    return rank, world, int(os.environ.get("LOCAL_RANK", "0")), world > 1 or "TORCHELASTIC_RUN_ID" in os.environ


def init_distributed(local, backend="nccl"):
    """One process per GPU; backend "nccl" IS RCCL on ROCm (gloo in the CPU test of this plumbing)."""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)


def timed_loop(step, steps, warmup, dist_on, device="cuda", telemetry=None, step_ms=None):
    """W untimed warm-up steps, then EXACTLY `steps` timed ones bracketed by a barrier + device synchronise on both sides;
    returns the MAX over ranks of the elapsed seconds (the job is as slow as its slowest rank).  `telemetry` (a
    _telemetry.Telemetry) samples clocks / power across warm-up and timed region; `step_ms` (a list) receives the HIP-event
    duration of every timed step (an event per step boundary on the current stream: no synchronisation added)."""
    import torch.distributed as dist
    sync = torch.cuda.synchronize if device == "cuda" else (lambda: None)
    if telemetry is not None:
        telemetry.start()
    for _ in range(warmup):
        step()
    if dist_on:
        dist.barrier()
    sync()
    marks = []
    if telemetry is not None:
        telemetry.mark("timed_begin")
    t0 = time.perf_counter()
    if step_ms is not None:
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[0].record()
    for _ in range(steps):
        step()
        if step_ms is not None:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    sync()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if telemetry is not None:
        telemetry.mark("timed_end")
        telemetry.stop()
    if step_ms is not None:
        step_ms.extend(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
    par = importlib.import_module("3dhumangan_amd.parallel")
    return par.max_over_ranks(dt, device=device if device == "cuda" else None)


def step_time_summary(step_ms):
    """min / median / first-5 / last-5 means of the per-step HIP-event times: a clock ramp (DVFS settling to the power
    budget) shows as first-5 < last-5."""
    if not step_ms:
        return None
    v = sorted(step_ms)
    k = min(5, len(step_ms))
    return dict(n=len(step_ms), min=v[0], median=v[len(v) // 2], max=v[-1], mean=sum(v) / len(v),
                first5_mean=sum(step_ms[:k]) / k, last5_mean=sum(step_ms[-k:]) / k,
                every=[round(x, 3) for x in (step_ms if len(step_ms) <= 40 else step_ms[::max(1, len(step_ms) // 40)])])


def timed_steps(G, cfg, z, cond, jitter, steps, warmup, dist_on, telemetry=None, step_ms=None):
    return timed_loop(lambda: G.forward(z, cond, jitter=jitter, **cfg), steps, warmup, dist_on, telemetry=telemetry,
                      step_ms=step_ms)


def discriminator_step_bench(a, rank, world, dist_on, dev):
    """BASELINE config 4, the discriminator half (the generator has no backward in this build): per rank `--batch` images of
    the reference-native 512x256 geometry (96x48 rays x 32 samples, hidden 256); a step = generator forward under no_grad
    (HIP kernels) + UNetDiscriminator forward on real and fake + R1 double backward + the all-gather of the R1 statistics +
    gradient all-reduce + Adam.  Prints its own JSON line."""
    trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
    disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
    G, cfg = build_generator(a.config, (512, 256), (96, 48), 32, dev)
    z, cond, jitter = make_inputs(cfg, a.batch, dev, seed=1234 + rank)
    torch.manual_seed(99)
    D = disc.UNetDiscriminator(**{k: v for k, v in cfg.items() if k != "neural_field_cls"}).to(dev)
    opt = torch.optim.Adam(D.parameters(), lr=cfg.get("disc_lr", 2e-4), betas=(0.0, 0.9))
    g = torch.Generator().manual_seed(7 + rank)
    real = torch.randn(a.batch, 3, 512, 256, generator=g).clamp(-1, 1).to(dev)
    gt = torch.randint(0, max(1, cfg.get("label_dim", 1)), (a.batch, 512, 256), generator=g).to(dev)
    # every loss term on (the shipped configs switch them per training phase): logistic GAN + segmentation head + R1
    meta = dict(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, label_dim=cfg.get("label_dim", 0))
    last = {}

    def step():
        with torch.no_grad():
            fake = G.forward(z, cond, jitter=jitter, **cfg)["rgbs"]
        last.update(trainers.discriminator_step(D, opt, real, fake, gt, meta, do_r1=True, distributed=dist_on,
                                                grad_clip=cfg.get("grad_clip", 10.0)))

    dt = timed_loop(step, a.steps, a.warmup, dist_on)
    if rank == 0:
        print(json.dumps({
            "metric": "discriminator-step images/sec at 512x256 (G forward + D fwd/bwd + R1)", "value": a.batch * world * a.steps / dt,
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "generator: x2 engines (f16 + fp6 cross terms); discriminator: split-bf16 convolution kernels (fp32 in / out)", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4, discriminator half: {a.config} 512x256, 96x48 rays x 32, batch {a.batch}/GPU; "
                                   "UNetDiscriminator 6 blocks; R1 every step; generator backward NOT included (not built)",
                       "global_batch": a.batch * world,
                       "parallelism": f"batch-sharded x{world}: RCCL all-gather of the R1 statistics + gradient all-reduce"},
            "loss": {k: float(v) for k, v in last.items()}}))


def train_step_bench(a, rank, world, dist_on, dev, emit=True):
    """BASELINE config 4: one whole adversarial iteration per step = discriminator step (generator forward in train mode under
    no_grad, D forward on real + fake, R1 double backward, all-gather of the R1 statistics, gradient all-reduce, Adam) +
    generator step (differentiable generator forward, D forward, backward through both, gradient all-reduce, Adam, EMA).
    Per rank `--batch` images of the reference-native 512x256 geometry (96x48 rays x 32 samples, hidden 256)."""
    trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
    disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
    ema_mod = importlib.import_module("3dhumangan_amd.lib.components.ema")
    G, cfg = build_generator(a.config, (512, 256), (96, 48), 32, dev)
    G.train()
    z, cond, jitter = make_inputs(cfg, a.batch, dev, seed=1234 + rank)
    torch.manual_seed(99)
    D = disc.UNetDiscriminator(**{k: v for k, v in cfg.items() if k != "neural_field_cls"}).to(dev)
    meta = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
    meta.update(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, gen_lr=5e-5, betas=(0.0, 0.9))
    opt_d = torch.optim.Adam(D.parameters(), lr=cfg.get("disc_lr", 2e-4), betas=(0.0, 0.9))
    opt_g = trainers.make_generator_optimizer(G, meta)
    ema = ema_mod.ExponentialMovingAverage(G.parameters(), decay=0.999)
    amp_dtype = {"none": None, "fp16": torch.float16, "bf16": torch.bfloat16}[a.amp]
    scaler = torch.amp.GradScaler("cuda") if a.amp == "fp16" else None      # ONE scaler for both steps, as base_trainer
    g = torch.Generator().manual_seed(7 + rank)
    real = torch.randn(a.batch, 3, 512, 256, generator=g).clamp(-1, 1).to(dev)
    gt = torch.randint(0, max(1, cfg.get("label_dim", 1)), (a.batch, 512, 256), generator=g).to(dev)
    last, ev = {}, {"d": [], "g": []}
    fwd = {k: v for k, v in cfg.items() if isinstance(k, str)}

    def step():
        e = {}

        def mark(name):
            e[name] = torch.cuda.Event(enable_timing=True)
            e[name].record()

        d, gs = trainers.adversarial_iteration(G, D, opt_d, opt_g, z, cond, real, gt, meta, generator_kwargs=dict(jitter=jitter),
                                               ema=ema, distributed=dist_on, grad_clip=cfg.get("grad_clip", 10.0),
                                               amp_dtype=amp_dtype, scaler=scaler, on_phase=mark)
        ev["d"].append((e["d"], e["g"]))
        ev["g"].append((e["g"], e["end"]))
        last.update({"d_" + k: v for k, v in d.items()})
        last.update({"g_" + k: v for k, v in gs.items()})

    dt = timed_loop(step, a.steps, a.warmup, dist_on)
    torch.cuda.synchronize()
    ms = {k: sum(s.elapsed_time(t) for s, t in v[-a.steps:]) / a.steps for k, v in ev.items()}
    line = None
    if rank == 0:
        line = ({
            "metric": "adversarial training iterations: images/sec at 512x256 (D step + G step)", "value": a.batch * world * a.steps / dt,
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp32" if a.amp == "none" else
                      f"AMP {a.amp} autocast + GradScaler (the reference's AMP mode): activations and their gradients travel as {a.amp} "
                      "through the hand-written convolution / weight-gradient / SPADE / sine kernels, weights and accumulation fp32") +
                     " (generator: split-bf16 matrix-core GEMMs / weight gradients + HIP activation / integration / SPADE kernels; discriminator: "
                     "hand-written split-bf16 convolution kernels, forward / backward-data / weight gradient / R1 double backward)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: {a.config} 512x256, 96x48 rays x 32, batch {a.batch}/GPU; UNetDiscriminator "
                                   "6 blocks; R1 every step; GAN + segmentation losses; Adam on both networks; EMA",
                       "g_step_discriminator_weight_gradients": trainers.g_step.G_STEP_D_GRADS,      # the reference computes them and zeroes them unread
                       "global_batch": a.batch * world,
                       "parallelism": f"batch-sharded x{world}: SyncBN moment all-reduces, RCCL all-gather of the R1 statistics, "
                                      "bucketed gradient all-reduce"},
            "stage_ms": {"discriminator_step": ms["d"], "generator_step": ms["g"]},
            "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
            "loss": {k: float(v) for k, v in last.items()}})
        if emit:
            print(json.dumps(line))
    return line


def kernel_rooflines(G, cfg, batch, stage_ms):
    """Algorithmic work of each HIP stage / measured HIP-event time.

    `achieved` is ALGORITHMIC flops (one multiply-add per weight and sample, what the reference computes) per second.
    On the split-operand engines every such product is issued as three f16/bf16 MFMA products, so the matrix pipe is
    3x busier than `frac` says: `mfma_pipe_util` = 3 * achieved / peak is the hardware utilisation."""
    Hd, F = cfg["hidden_dim"], cfg["feature_dim"]
    H, W = cfg["gen_height"], cfg["gen_width"]
    R, S = cfg["render_height"] * cfg["render_width"], cfg["num_steps"]
    n_mod = len(cfg["mod_blocks"]) if cfg["map3d_mode"] != "all" else cfg["synthesis_blocks"]
    pts, px = batch * R * S, batch * H * W
    out = {}

    def mfma_entry(flop, ms, engine, extra=None):
        """engine: the arithmetic of the kernel's contractions.  Matrix-pipe time issued per algorithmic f16-rate product:
        x3 = three f16/bf16 products; x2 = one f16 product + one block-scaled fp6 instruction per two k-steps that takes the
        time of one f16 instruction (both cross terms) = 1.5; f32 = the fp32 matrix instruction (its own peak)."""
        kind = "x2" if engine in ("f16x2", "f16x2t") else "x3" if engine.endswith("x3") or engine.endswith("x3t") else "f32"
        peak = MFMA_F32_PEAK_TF if kind == "f32" else MFMA_F16_PEAK_TF
        factor = {"x3": 3.0, "x2": 1.5, "f32": 1.0}[kind]
        label = {"x3": "split f16/bf16 x3: hi*hi + hi*lo + lo*hi, three 16-bit MFMA products (fp32-class)",
                 "x2": "x2: f16 hi*hi + one block-scaled fp6 (e2m3) MFMA for both cross terms (1e-3-class, measured 2e-4)",
                 "f32": "fp32 MFMA"}[kind]
        ach = flop / ms / 1e9
        e = dict(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, ms=ms, flop=flop, engine=f"{engine}: {label}",
                 mfma_issue_factor=factor, mfma_pipe_util=factor * ach / peak, frac_of_fp32_mfma_peak=ach / MFMA_F32_PEAK_TF)
        if extra:
            e.update(extra)
        return e

    if "render_fused" in stage_ms:
        fl = 2.0 * (7 * Hd * Hd + 41 * Hd) * pts
        out["h3d_render_fused"] = mfma_entry(fl, stage_ms["render_fused"][0], G.neural_field.precision)
    if "synthesis" in stage_ms:
        executed = 2.0 * (18 * Hd * Hd + 2 * n_mod * 128 * 2 * Hd + 6 * 3 * Hd) * px      # after the exact folding
        reference_form = 2.0 * (18 * Hd * Hd + 6932 * Hd) * px                              # SURVEY 8(d) figure
        ms = stage_ms["synthesis"][0]
        out["h3d_synthesis"] = mfma_entry(executed, ms, G.synthesis_plan(next(G.parameters()).device).engine,
                                          dict(reference_formulation_TFLOPs=reference_form / ms / 1e9,
                                               reference_formulation_frac=reference_form / ms / 1e9 / MFMA_F16_PEAK_TF))
    if "geo_features" in stage_ms:
        ms = stage_ms["geo_features"][0]
        fused_geo = bool(getattr(G, "fuse_geo", False)) and G.neural_field.render_geo_supported(S)
        if fused_geo:       # A4 inside the render: this stage is the search alone (points in, one int32 index out, mesh once)
            by, name = pts * (3 + 1) * 4.0 + batch * 6890 * 3 * 4.0, "h3d_nearest_vertex"
        else:               # points in, features out, mesh + transforms once
            by, name = pts * (3 + 31) * 4.0 + batch * 6890 * (3 + 3 + 16) * 4.0, "h3d_geo_features"
        pruned = os.environ.get("H3D_NN_PRUNE", "1") != "0"
        out[name] = dict(bound="valu+mfma (6890 point-vertex pairs per sample); HBM figures for reference",
                         entry_points=("h3d_mesh_sort + " + name + "_sorted (chunk pruning on the Morton-sorted mesh)") if pruned else name,
                         ms=ms, point_vertex_pairs_per_s=pts * 6890 / ms * 1e3, bytes=by,
                         hbm_achieved_GBs=by / ms / 1e6, hbm_frac=by / ms / 1e6 / HBM_PEAK_GBS,
                         note="stage time includes the [V,24]x[24,16] blended-transform GEMM of the pose (library, per batch) and the "
                              "Morton sort of the pose's mesh; point_vertex_pairs_per_s counts ALL pairs of the brute-force search the "
                              "reference runs -- the sorted kernel skips ~4/5 of the 64-vertex chunks by bounding sphere (exactly: "
                              "same indices)")
    return out


def ray_integrate_roofline(cfg, batch, iters=10):
    """The stand-alone A6 kernel (drop-in for vr.ray_integration) on this workload's field tensor: the HBM-bound
    kernel the north_star's >=40% target is evaluated on."""
    R, S, C = cfg["render_height"] * cfg["render_width"], cfg["num_steps"], cfg["feature_dim"] + 3
    nb = batch
    while nb > 1 and nb * R * S * (C + 1) * 4 > 12e9:
        nb //= 2
    field = torch.randn(nb, R, S, C + 1, device="cuda")
    z = torch.sort(torch.rand(nb, R, S, device="cuda") + 11, dim=2).values.contiguous()
    feats = torch.empty(nb, R, C, device="cuda")
    depth = torch.empty(nb, R, device="cuda")
    wts = torch.empty(nb, R, S, device="cuda")
    L = importlib.import_module("3dhumangan_amd._lib")
    lib = L.load()
    st = L.stream_handle()

    def run():      # straight through the C ABI, no per-call allocation: the events bracket kernels only
        L.check(lib.h3d_ray_integrate(L.ptr(field), L.ptr(z), None, L.ptr(feats), L.ptr(depth), L.ptr(wts), nb * R, S, C,
                                      0, 1, 1, st), "h3d_ray_integrate")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    by = 4.0 * nb * R * (S * (C + 1) + S + C + 1 + S)
    return dict(bound="hbm", achieved=by / ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s", frac=by / ms / 1e6 / HBM_PEAK_GBS,
                traffic=None, ms=ms, bytes=by, batch=nb)


def load_traffic(workload_key):
    """({kernel: HBM bytes per launch}, file, {kernel: matrix-pipe busy fraction from the SQ counter pass, when the file has it},
    date the file states) from the newest profiles/*_hbm_traffic.json measured on this workload.  These are counters of a BUILDER
    run (separate rocprofv3 --pmc passes, tools/profile_round.sh), never of the run that prints them: the line says so."""
    import glob
    best, src, busy, when = {}, None, {}, None
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload_key:
            best = {k: v["bytes_per_launch"] for k, v in d.get("kernels", {}).items()}
            busy = {k: v["mfma_busy_frac"] for k, v in d.get("kernels", {}).items() if "mfma_busy_frac" in v}
            src, when = "profiles/" + os.path.basename(f), d.get("date")
    return best, src, busy, when


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_run(cfg, sd, batch, seed, runs):
    """1 warm-up + `runs` timed oracle forwards of `batch` images -> [seconds]."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import h3d_oracle as O
    synthetic = importlib.import_module("3dhumangan_amd.synthetic")
    g = torch.Generator().manual_seed(seed)
    cond = synthetic.make_conditions(batch, 6890, seed=seed % 1000)
    z = torch.randn(batch, cfg["latent_dim"], generator=g)
    jit = torch.rand(batch, cfg["render_height"] * cfg["render_width"], cfg["num_steps"], 1, generator=g)
    times = []
    with torch.no_grad():
        for i in range(runs + 1):
            t0 = time.perf_counter()
            O.generator_forward(sd, cfg, z, cond, jit, None)
            if i:
                times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(cfg, sd, seed=1234, shrink=1, runs=1):
    """The CPU oracle (a port of the reference path, pinned to the reference by tests/golden) on the host cores.
    Bounded sample of the SAME workload: ONE image at FULL size (shrink = 1: the output pixels, rays, samples per ray, widths
    and weights of the timed GPU workload; round 6 -- earlier rounds timed a half-size image and multiplied by four), 1 warm-up
    + `runs` timed runs, median.  `shrink` > 1 (development) times a 1/shrink-size image and scales by shrink^2."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))      # more threads than this only adds contention
    ocfg = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
    for k in ("gen_height", "gen_width", "render_height", "render_width"):
        ocfg[k] = max(1, cfg[k] // shrink)
    times = sorted(_oracle_run(ocfg, sd, 1, seed, runs))
    dt = times[len(times) // 2]
    full = dt * shrink * shrink
    scale = "" if shrink == 1 else f" x{shrink * shrink} = {full:.0f} s/full-size image (work linear in rays and pixels)"
    return dict(value=1.0 / full, unit="images/s", cores=torch.get_num_threads(), kind="port", cpu=_cpu_model(),
                host_cores=os.cpu_count(), torch=torch.__version__, runs=[round(t, 2) for t in times],
                sample=f"1 image at {'FULL' if shrink == 1 else f'1/{shrink} linear'} size ({ocfg['gen_height']}x{ocfg['gen_width']} px, "
                       f"{ocfg['render_height']}x{ocfg['render_width']} rays x {cfg['num_steps']}), 1 warm-up + {runs} timed, median {dt:.1f} s"
                       f"{scale}; {torch.get_num_threads()} of {os.cpu_count()} host threads",
                note="pure-PyTorch CPU oracle (a port of the reference path, pinned to it by tests/golden), brute-force nearest-vertex "
                     "search")


def cpu_baseline_cfg1(runs=1):
    """BASELINE config 1 on the host cores, FULL size: MAP3DBN (hidden 384), one 256x128 image from 64x32 rays x 32
    samples (the reference-native '256^2'); 1 warm-up + `runs` timed runs of the oracle, images/s = 1 / median."""
    configs = importlib.import_module("3dhumangan_amd.configs")
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
    cfg = {k: v for k, v in configs.MAP3DBN.items() if isinstance(k, str)}
    cfg.update(dataset_length=4, nerf_noise=0, last_back=cfg["eval_last_back"])
    torch.manual_seed(1234)
    G = gens.Map3DGenerator(**dict(cfg, neural_field_cls=impl.COORDCONCATSIREN)).eval()
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    cfg.pop("neural_field_cls", None)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    times = sorted(_oracle_run(cfg, sd, 1, 1234, runs))
    med = times[len(times) // 2]
    return dict(value=1.0 / med, unit="images/s", cores=torch.get_num_threads(), kind="port", runs=[round(t, 2) for t in times],
                sample=f"MAP3DBN 256x128, 64x32 rays x 32, batch 1, full size: 1 warm-up + {runs} timed runs")


def cpu_baseline_cfg2(batch=8, shrink=2, runs=1):
    """BASELINE config 2 on the host cores ("cfg 2 (B=8)"): MAP3DBN (hidden 384), batch 8 of the reference-native
    256x128 image from 64x32 rays x 32 samples, at 1/shrink linear size to bound the time (work is linear in rays and pixels):
    1 warm-up + `runs` timed oracle forwards of the whole batch, images/s = batch / (median * shrink^2)."""
    configs = importlib.import_module("3dhumangan_amd.configs")
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
    cfg = {k: v for k, v in configs.MAP3DBN.items() if isinstance(k, str)}
    cfg.update(dataset_length=4, nerf_noise=0, last_back=cfg["eval_last_back"])
    torch.manual_seed(1234)
    G = gens.Map3DGenerator(**dict(cfg, neural_field_cls=impl.COORDCONCATSIREN)).eval()
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    cfg.pop("neural_field_cls", None)
    for k in ("gen_height", "gen_width", "render_height", "render_width"):
        cfg[k] = max(1, cfg[k] // shrink)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    times = sorted(_oracle_run(cfg, sd, batch, 1234, runs))
    med = times[len(times) // 2]
    return dict(value=batch / (med * shrink * shrink), unit="images/s", cores=torch.get_num_threads(), kind="port",
                runs=[round(t, 2) for t in times], batch=batch,
                sample=f"MAP3DBN batch {batch} at 1/{shrink} linear size ({cfg['gen_height']}x{cfg['gen_width']} px, "
                       f"{cfg['render_height']}x{cfg['render_width']} rays x 32): 1 warm-up + {runs} timed runs of the whole batch, "
                       f"median {med:.1f} s -> {med * shrink * shrink:.0f} s per full-size batch")


def check_cells(cfg, item, frac=0.05, seed=5, bright=()):
    """Low-resolution cells (cy, cx) of the pixels self_check compares for batch item `item`: a contiguous K x K patch of cells
    (neighbouring pixels share their rays: (K + 1)^2 rays for K^2 cells) at a per-item seeded position, K the smallest size whose
    pixels are >= `frac` of the image, plus the two corner cells, the centre cell and the cells `bright` (those holding the largest
    |value| of each channel: there the oracle's maximum over the checked pixels IS the image's scale)."""
    H, W, Hr, Wr = cfg["gen_height"], cfg["gen_width"], cfg["render_height"], cfg["render_width"]
    per_cell = (H / Hr) * (W / Wr)
    K = 1
    while K < min(Hr, Wr) - 1 and K * K * per_cell < frac * H * W:
        K += 1
    g = torch.Generator().manual_seed(seed * 1000 + item)
    cy = int(torch.randint(0, max(1, Hr - K), (1,), generator=g))
    cx = int(torch.randint(0, max(1, Wr - K), (1,), generator=g))
    cells = [(0, 0), (Hr - 1, Wr - 1), (Hr // 2, Wr // 2)] + list(bright) + [(cy + i, cx + j) for i in range(K) for j in range(K)]
    return cells, K


def brightest_cells(img, render_hw):
    """img [3, H, W] -> the low-resolution cells (upper-left bilinear tap, as pixels_of_cells counts them) of the pixel with the
    largest |value| of each channel.  Only the LOCATION comes from the image under test; the scale the errors are divided by is the
    oracle's value there."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import h3d_oracle as O
    H, W = img.shape[1:]
    y0, _, _ = O._resize_axis(render_hw[0], H)
    x0, _, _ = O._resize_axis(render_hw[1], W)
    out = []
    for c in range(img.shape[0]):
        p = int(img[c].abs().argmax())
        out.append((int(y0[p // W]), int(x0[p % W])))
    return out


def self_check(G, cfg, z, cond, jitter, items, frac=0.05, seed=5):
    """Correctness of what was timed: one more forward of the SAME batch, compared with the CPU oracle restricted to a
    subset of pixels / rays (oracle/h3d_oracle.py: generator_forward_subset) for every batch item in `items`: a contiguous
    patch of >= `frac` (5 %) of the item's pixels at a per-item position plus corner / centre cells and the cells holding each
    channel's largest |value| (check_cells, brightest_cells).

    Norms (round 6): `max_rel_err` = per-channel max |difference| over the checked pixels / per-channel max |ORACLE| over the
    checked pixels -- the scale comes from the oracle, not from the output under test, and the pass gate uses it; the checked set
    contains the image's brightest cells, so that maximum is the image's scale and not that of a possibly dim patch
    (`max_rel_err_patch_norm`: the patch alone, for the record);
    `max_rel_err_image_norm` = the same differences / per-channel max over the WHOLE output image (what rounds 4-5 printed as
    max_rel_err), reported beside it.

    Rays on the reference's last-sample discontinuity (delta = 1e9 for the last sample, lib/generators/volume_rendering.py:21:
    its alpha is 0 or 1 by the SIGN of the density, and with white_back the background term flips by the whole remaining
    transmittance) are left out of the tolerance ONLY when the oracle itself says the ray is ill-conditioned: its last-sample
    density lies within the parity tolerance of zero (|sigma_last| <= 1e-3 * max|sigma| of the item, `ill_conditioned_rays`).
    A ray with a large error whose oracle density is NOT near zero counts as the error it is.  Only the bilinear footprint of
    such a ray (the output pixels with that ray among their four taps) is masked in the image comparison; the number of
    excluded rays is capped (`max_excluded`), beyond it the check fails."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import h3d_oracle as O
    # torchrun exports OMP_NUM_THREADS=1 to its ranks when N > 1: the oracle leg of rank 0 would run on one thread
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    out = G.forward(z, cond, jitter=jitter, **cfg)
    rgb, ren = out["rgbs"].cpu(), out["rgbs_render"].cpu()
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    ocfg = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
    Hr, Wr = cfg["render_height"], cfg["render_width"]
    worst, worst_r, worst_img, rays, per_item, per_item_img, excluded, n_rays, n_pix, flips = 0.0, 0.0, 0.0, 0, [], [], 0, 0, 0, 0
    worst_patch, flips_excl = 0.0, 0
    zc, jc = z.cpu(), jitter.cpu()
    K = 0
    # the band of last-sample densities the ORACLE calls ill-conditioned for the arithmetic under test: 1e-3 of the item's largest
    # density for the plain x2 render (its density error is ~1e-4), 1e-4 for fp32-class arithmetic -- the three-product / fp32
    # engines, and the x2 render with its last-sample refinement (round 6: the rays inside 1e-3 are redone on three products)
    nf = G.neural_field
    ill_rel = 1e-3 if (nf.precision in ("f16x2", "f16x2t", "f16x1t") and not (nf.precision == "f16x2" and getattr(nf, "refine_last_sample", False)
                                                                              and getattr(G, "fuse_geo", False))) else 1e-4
    t0 = time.perf_counter()
    for i in items:
        bright = brightest_cells(rgb[i], (Hr, Wr))
        cells, K = check_cells(cfg, i, frac, seed, bright)
        pix = O.pixels_of_cells(cells, (cfg["gen_height"], cfg["gen_width"]), (Hr, Wr))
        in_patch = torch.isin(pix, O.pixels_of_cells(cells[3 + len(bright):], (cfg["gen_height"], cfg["gen_width"]), (Hr, Wr)))
        ci = {k: v[i:i + 1].cpu() for k, v in cond.items()}
        ref = O.generator_forward_subset(sd, ocfg, zc[i:i + 1], ci, jc[i:i + 1], pix)
        got = rgb[i:i + 1].flatten(2)[:, :, pix]
        got_r = ren[i:i + 1].flatten(2)[:, :, ref["ray_subset"]]
        dr = got_r - ref["rgbs_render"]
        ill = ill_conditioned_rays(ref["sigma"], ill_rel)[0]              # [Rs] bool, from the ORACLE's densities alone
        keep_ray = ~ill
        keep_px = ~ill[ref["taps"]].any(0)                                # [P]: pixels none of whose four taps is such a ray
        excluded += int(ill.sum())
        n_rays += int(ill.numel())
        n_pix += int(len(pix))
        flips += int((discontinuity_rays(dr)[0] & keep_ray).sum())        # a flip on a ray the oracle calls well-conditioned: counted as error below
        flips_excl += int((discontinuity_rays(dr)[0] & ill).sum())        # ... and how many of the excluded rays actually flipped
        w_i, w_img = 0.0, 0.0
        for c in range(3):
            d = ((got[:, c] - ref["rgbs"][:, c]).abs() * keep_px).max()
            w_i = max(w_i, float(d / ref["rgbs"][:, c].abs().max()))      # scale from the ORACLE (checked pixels, brightest cells included)
            worst_patch = max(worst_patch, float(((got[:, c] - ref["rgbs"][:, c]).abs() * keep_px * in_patch).max()
                                                 / (ref["rgbs"][:, c].abs() * in_patch).max()))
            w_img = max(w_img, float(d / rgb[i, c].abs().max()))          # scale of the whole output image (rounds 4-5)
            worst_r = max(worst_r, float((dr[:, c].abs() * keep_ray).max() / ref["rgbs_render"][:, c].abs().max()))
        rays = len(ref["ray_subset"])
        per_item.append(w_i)
        per_item_img.append(w_img)
        worst, worst_img = max(worst, w_i), max(worst_img, w_img)
    plan = G.synthesis_plan(z.device)
    max_excluded = max(1, int(5e-4 * n_rays + 0.5))
    x2 = plan.engine in ("f16x2", "f16x2t")
    mon = plan.x2_monitor_errors() if (plan.engine == "f16x2" and plan.x2_monitor) else None
    return dict(max_rel_err=worst, max_rel_err_render=worst_r, max_rel_err_image_norm=worst_img, max_rel_err_patch_norm=worst_patch,
                tolerance=1e-3,
                norm="per-channel max |difference| over the checked pixels / per-channel max |oracle| over the checked pixels, which "
                     "include each channel's brightest cell (max_rel_err_image_norm: / per-channel max over the whole output image; "
                     "max_rel_err_patch_norm: the >= 5 % patch alone, difference and scale)",
                ok=bool(worst < 1e-3 and worst_r < 1e-3 and excluded <= max_excluded),
                batch_items=list(items), pixels_per_item=int(n_pix // max(1, len(items))), rays_per_item=int(rays),
                pixel_fraction=n_pix / max(1, len(items)) / (cfg["gen_height"] * cfg["gen_width"]), patch_cells=K * K,
                per_item_max_rel_err=[round(e, 7) for e in per_item], per_item_max_rel_err_image_norm=[round(e, 7) for e in per_item_img],
                rays_excluded_as_ill_conditioned_in_the_oracle=excluded, max_excluded=max_excluded, rays_checked=n_rays,
                ill_conditioned_band=ill_rel,
                refined_units=(nf.refined_units().tolist() if getattr(nf, "refined_units", None) and nf.refined_units() is not None else None),
                discontinuity_signatures_on_well_conditioned_rays=flips, discontinuity_signatures_on_excluded_rays=flips_excl,
                oracle_seconds=time.perf_counter() - t0,
                synthesis_engine=plan.engine, field_engine=G.neural_field.precision,
                x2_fallback_items=plan.x2_fallback_items() if x2 else None,
                x2_monitor=(dict(tolerance=plan.x2_monitor_tol, max_sampled_err=float(mon.max()),
                                 tiles_per_image=plan.monitor_tile_count(cfg["gen_height"], cfg["gen_width"])) if mon is not None else None),
                against="CPU oracle on a pixel/ray subset (per-channel max-norm); oracle pinned to the reference's vectors")


def ill_conditioned_rays(sigma, rel=1e-3):
    """sigma [B, Rs, S] = the ORACLE's densities (noise added) of the checked rays -> bool [B, Rs]: rays whose LAST sample's
    density is within the parity tolerance of zero (|sigma_last| <= rel * max|sigma|).  The reference gives the last sample
    delta = 1e9 (lib/generators/volume_rendering.py:21), so alpha_last = 1 - exp(-1e9 relu(sigma_last)) is 0 or 1 by the sign of
    sigma_last: a perturbation of sigma inside the tolerance flips the ray's background term by its whole remaining
    transmittance.  The condition looks at the oracle only -- not at the difference being judged."""
    return sigma[:, :, -1].abs() <= rel * sigma.abs().amax(dim=(1, 2), keepdim=True)[:, :, 0]


def discontinuity_rays(dr):
    """dr [B, 3, R] = rendered colour minus reference -> bool [B, R]: rays whose difference has the SIGNATURE of the last-sample
    discontinuity (far above the tolerance and the same in all three channels).  A diagnostic only -- a compositing bug has the
    same signature -- so nothing is excluded on it: self_check excludes on `ill_conditioned_rays` (the oracle's own densities)."""
    return (dr.abs().amax(1) > 1e-2) & ((dr.amax(1) - dr.amin(1)) < 1e-3)


def op_rooflines():
    """HBM rooflines of the stand-alone HBM-bound ops: algorithmic bytes (inputs + outputs once) / time.  The kernels
    are called straight through the C ABI on PREALLOCATED outputs (as ray_integrate_roofline does): the HIP events bracket kernel
    launches only, no allocation, no Python wrapper."""
    import ctypes
    L = importlib.import_module("3dhumangan_amd._lib")
    uf = importlib.import_module("3dhumangan_amd.lib.components.ops.upfirdn2d")
    lib, st = L.load(), L.stream_handle()

    def timeit(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    def entry(by, ms, shape):
        return dict(bound="hbm", achieved=by / ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s", frac=by / ms / 1e6 / HBM_PEAK_GBS,
                    ms=ms, bytes=by, shape=shape, note="C ABI on preallocated outputs: kernel time only")

    out = {}
    x = torch.empty(8, 64, 512, 512, device="cuda")
    ms = timeit(lambda: x.fill_(1.0))
    out["hbm_write_only_reference"] = dict(bound="hbm", achieved=x.numel() * 4.0 / ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s",
                                           frac=x.numel() * 4.0 / ms / 1e6 / HBM_PEAK_GBS, ms=ms, bytes=x.numel() * 4.0,
                                           note="a plain fill of a 537 MB tensor: what a write-only stream reaches on this box (context for "
                                                "the write-dominated resampling kernels below; the 8 TB/s peak is read + write)")
    del x
    # P1 bias_act: lrelu (activation index 3, alpha 0.2, gain sqrt 2, no clamp), bias along dim 1 of [8,256,512,256] f32
    x = torch.randn(8, 256, 512, 256, device="cuda")
    bvec = torch.randn(256, device="cuda")
    y = torch.empty_like(x)

    def run_bias_act():
        L.check(lib.h3d_bias_act(L.ptr(x), L.ptr(bvec), L.ptr(y), x.numel(), 0, 256, x.stride(1), 3, 0.2, 2.0 ** 0.5, -1.0, st),
                "h3d_bias_act")

    out["h3d_bias_act"] = entry(2.0 * x.numel() * 4, timeit(run_bias_act), "lrelu [8,256,512,256] f32")
    del x, y
    # P2 upfirdn2d: 2x up with the 4x4 [1,3,3,1] filter (one pass), padding (2,1,2,1), gain 4
    x = torch.randn(8, 64, 256, 256, device="cuda")
    f = uf.setup_filter([1, 3, 3, 1], device="cuda", separable=False).contiguous()
    y = torch.empty(8, 64, 512, 512, device="cuda")
    xs, ys = (ctypes.c_int64 * 4)(*x.stride()), (ctypes.c_int64 * 4)(*y.stride())

    def run_upfirdn():
        L.check(lib.h3d_upfirdn2d(L.ptr(x), L.ptr(f), L.ptr(y), 0, 8, 64, 256, 256, xs, 4, 4, 512, 512, ys, 2, 2, 1, 1, 2, 2, 0, 4.0, st),
                "h3d_upfirdn2d")

    out["h3d_upfirdn2d"] = entry((x.numel() + y.numel()) * 4.0, timeit(run_upfirdn),
                                 "2x up, 4x4 [1,3,3,1] filter, [8,64,256,256] -> [8,64,512,512] f32")
    ref = uf.upfirdn2d(x, f, up=2, padding=(2, 1, 2, 1), gain=4)                  # the wrapper's result: same launch parameters
    out["h3d_upfirdn2d"]["matches_wrapper"] = bool(torch.equal(ref, y))
    del x, y, ref
    # A7 bilinear resize [4,256,96,96] -> [4,256,512,512]
    x = torch.randn(4, 256, 96, 96, device="cuda")
    y = torch.empty(4, 256, 512, 512, device="cuda")

    def run_bilinear():
        L.check(lib.h3d_bilinear_resize(L.ptr(x), L.ptr(y), 4, 256, 96, 96, 512, 512, st), "h3d_bilinear_resize")

    out["h3d_bilinear_resize"] = entry((x.numel() + y.numel()) * 4.0, timeit(run_bilinear), "[4,256,96,96] -> [4,256,512,512] f32")
    del x, y
    out["conv_x3_by_shape"] = conv_rooflines(timeit)
    return out


def conv_rooflines(timeit):
    """Per-shape table of the discriminator's / dense layers' matrix-core kernels (the conv_x3 roofline must be
    recomputable): algorithmic flops 2 B H W Ci Co k^2 and bytes (input + output once, weights negligible) against measured time,
    for fp32 activations and for the AMP tier's f16 activations.  `mfma_pipe_util` = matrix instructions issued per algorithmic
    product x frac: three bf16 products for fp32 operands (x3); for f16 operands two F16 products in the convolution (the
    activation is exact in one plane, the weights travel as f16 hi + lo) and ONE in the weight gradient (both operands exact).
    `hbm_frac` = bytes / time / 8 TB/s.  Times include the per-call weight packing launch (cached per weight version)."""
    conv = importlib.import_module("3dhumangan_amd.lib.components.ops.conv")
    rows = []
    shapes = [(4, 512, 256, 128, 128, 3), (4, 256, 128, 128, 256, 3), (4, 256, 128, 256, 256, 3), (4, 128, 64, 256, 512, 3),
              (4, 128, 64, 512, 512, 3), (4, 64, 32, 512, 512, 3), (4, 512, 256, 256, 128, 3), (4, 512, 256, 128, 128, 1),
              (1, 1, 524288, 256, 256, 1)]
    for B, H, W, ci, co, k in shapes:
        w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
        for dt in (torch.float32, torch.float16):
            x = torch.randn(B, ci, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
            g = torch.randn(B, co, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
            flop = 2.0 * B * H * W * ci * co * k * k
            esz = 4 if dt == torch.float32 else 2
            cases = [("forward", lambda: conv._run_conv(x, w), B * H * W * (ci + co) * esz),
                     ("weight_gradient", lambda: conv._run_wgrad(x, g, k), B * H * W * (ci + co) * esz)]
            if k == 1:      # the library's GEMM on the same rows (hipBLASLt: what H3D_AMP_LINEAR=library / H3D_LINEAR=library run)
                x2 = x.permute(0, 2, 3, 1).reshape(-1, ci)
                w2 = w.reshape(co, ci).to(dt)
                cases.append(("forward_library_gemm", lambda: torch.nn.functional.linear(x2, w2), B * H * W * (ci + co) * esz))
            for name, fn, by in cases:
                ms = timeit(fn, iters=5)
                ach = flop / ms / 1e9
                planes = getattr(conv, "AMP_WEIGHT_PLANES", 2)
                issue = (1.0 if name == "forward_library_gemm" else 3.0 if dt == torch.float32
                         else float(planes) if name == "forward" else 1.0)
                rows.append(dict(kernel="h3d_conv_x3" if name == "forward" else "hipBLASLt" if name == "forward_library_gemm" else "h3d_conv_wgrad_x3", pass_=name,
                                 shape=f"B{B} {H}x{W} {ci}->{co} k{k}", activations="f32" if dt == torch.float32 else "f16",
                                 ms=ms, flop=flop, bytes=by, achieved_TFLOPs=ach, frac=ach / MFMA_F16_PEAK_TF,
                                 mfma_issue_factor=issue, mfma_pipe_util=issue * ach / MFMA_F16_PEAK_TF, hbm_GBs=by / ms / 1e6,
                                 hbm_frac=by / ms / 1e6 / HBM_PEAK_GBS))
            del x, g
    return rows


LINE_LIMIT = 4096             # the driver parses the LAST stdout line; round 4's 25.6 KB line came back unparsed


def _finite(x):
    """JSON-strict scalars: NaN / Infinity -> None (json.dumps would print the bare words, which strict parsers reject)."""
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return None
    return x


def _sanitize(o):
    if isinstance(o, dict):
        return {str(k): _sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sanitize(v) for v in o]
    return _finite(o)


def _r(x, n=4):
    return None if x is None else _finite(round(float(x), n))


def compact_line(out):
    """The ONE stdout line: the contract keys only, scalars rounded, <= LINE_LIMIT bytes, no array longer than 16, strict JSON.
    Everything else (per-kernel table, telemetry series, per-shape convolution table, per-step times, per-item errors) goes to
    bench_detail.json (write_detail)."""
    roof = out.get("roofline") or {}
    hbm = out.get("roofline_hbm_kernel") or {}
    cpu = out.get("cpu_baseline")
    chk = out.get("checked")
    ex = out.get("extra") or {}

    def ips(key):
        v = ex.get(key)
        return _r(v.get("images_per_s"), 2) if isinstance(v, dict) and "images_per_s" in v else None

    extra = {
        "native_512x256_images_per_s": _r(ex.get("native_512x256_images_per_s"), 2),
        "cfg2_256sq_b8_images_per_s": ips("cfg2_MAP3DBN_256x256_64x64rays_s32"),
        "cfg3L_hidden420_images_per_s": ips("cfg3L_MAP3DBN512L_512x512_96x96rays_s64"),
        "cfg5_1024sq_s128_b4_images_per_s": ips("cfg5_MAP3DBN512_1024x1024_192x192rays_s128"),
        "x3_engines_images_per_s": ips("headline_workload_on_x3_engines"),
        "cfg4_trainstep_b4_fp32_ms": _r((ex.get("cfg4_trainstep_b4") or {}).get("ms_per_iteration"), 2),
        "cfg4_trainstep_b4_amp_fp16_ms": _r((ex.get("cfg4_trainstep_b4_amp_fp16") or {}).get("ms_per_iteration"), 2),
        "cpu_cfg1_images_per_s": _r((ex.get("cpu_baseline_cfg1") or {}).get("value"), 4),
        "cpu_cfg2_b8_images_per_s": _r((ex.get("cpu_baseline_cfg2_b8") or {}).get("value"), 4),
        "joules_per_image": _r(((out.get("telemetry") or {}).get("timed") or {}).get("joules_per_image"), 3),
    }
    line = {
        "metric": out["metric"], "value": _r(out["value"], 3), "unit": out["unit"], "n_gpus": out["n_gpus"],
        "steps": out["steps"], "warmup": out["warmup"], "ms_per_step": _r(out["ms_per_step"], 4),
        "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": out["vs_baseline"],
        "dtype": out["dtype"], "data": out["data"], "config": out["config"],
        "roofline": {"kernel": roof.get("kernel"), "bound": roof.get("bound"), "achieved": _r(roof.get("achieved"), 2),
                     "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": _r(roof.get("frac"), 4),
                     "traffic": roof.get("traffic"), "traffic_source": roof.get("traffic_source_short"), "ms": _r(roof.get("ms"), 4),
                     "basis": "achieved / frac: SURVEY 8(d) reference-formulation flops per launch; *_executed: flops issued after the exact folding",
                     "achieved_executed": _r(roof.get("achieved_executed"), 2), "frac_executed": _r(roof.get("frac_executed"), 4),
                     "mfma_busy_pmc": roof.get("mfma_busy_pmc")} if roof else None,
        "roofline_hbm_kernel": {"kernel": hbm.get("kernel"), "bound": hbm.get("bound"), "achieved": _r(hbm.get("achieved"), 1),
                                "peak": hbm.get("peak"), "unit": hbm.get("unit"), "frac": _r(hbm.get("frac"), 4),
                                "traffic": hbm.get("traffic")} if hbm else None,
        "cpu_baseline": {"value": _r(cpu["value"], 5), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                         "sample": cpu["sample"][:200]} if cpu else None,
        "checked": {"max_rel_err": _r(chk["max_rel_err"], 7), "max_rel_err_render": _r(chk["max_rel_err_render"], 7),
                    "max_rel_err_image_norm": _r(chk.get("max_rel_err_image_norm"), 7),
                    "max_rel_err_patch_norm": _r(chk.get("max_rel_err_patch_norm"), 7),
                    "norm": "max|diff| / max|oracle| per channel over the checked pixels incl. each channel's brightest cell (image_norm: / max of the whole output image; patch_norm: the 5 % patch alone)",
                    "ok": chk["ok"], "tolerance": chk["tolerance"], "items": len(chk["batch_items"]),
                    "pixel_fraction_per_item": _r(chk.get("pixel_fraction"), 4), "rays_per_item": chk.get("rays_per_item"),
                    "rays_excluded": chk["rays_excluded_as_ill_conditioned_in_the_oracle"], "rays_checked": chk.get("rays_checked"),
                    "ill_band": chk.get("ill_conditioned_band"), "refined_units": sum(chk.get("refined_units") or []) if chk.get("refined_units") is not None else None,
                    "x2_monitor_err": _r((chk.get("x2_monitor") or {}).get("max_sampled_err"), 7),
                    "x2_monitor_tol": (chk.get("x2_monitor") or {}).get("tolerance"),
                    "x2_fallback_items": chk.get("x2_fallback_items")} if chk else None,
        "stage_ms": {k: _r(v, 3) for k, v in list((out.get("stage_ms") or {}).items())[:8]},
        "extra": {k: v for k, v in extra.items() if v is not None},
        "detail": "bench_detail.json",
    }
    def clip(o, n=400):                              # no string of the line is longer than n characters
        if isinstance(o, dict):
            return {k: clip(v, n) for k, v in o.items()}
        return o[:n] if isinstance(o, str) else o

    line = clip(line)
    text = json.dumps(_sanitize(line), allow_nan=False, separators=(", ", ": "))
    if len(text) >= LINE_LIMIT:                      # never lose the headline to a long string: drop the optional parts
        for k in ("extra", "stage_ms", "roofline_hbm_kernel"):
            line.pop(k, None)
            text = json.dumps(_sanitize(line), allow_nan=False, separators=(", ", ": "))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT and "\n" not in text, len(text)
    return text


def write_detail(out, name="bench_detail.json"):
    """The full record (everything the compact line leaves out) next to the script and, when it exists, under gpurun_out/."""
    text = json.dumps(_sanitize(out), allow_nan=False, indent=1)
    written = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    f.write(text)
                written.append(os.path.join(d, name))
            except OSError:
                pass
    return written


def respawn_under_torchrun(a, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-exec this script as N ranks, one per GPU, the way
    the driver's contract command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...).  Returns only when no re-exec is needed."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but this node shows {have} GPU(s)")
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def check_world(a, world, backend_world=None):
    """n_gpus == --gpus == WORLD_SIZE == the process group's size, or fail loudly (a mislabelled scaling point is worse than
    none)."""
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {a.gpus} "
                         "(or run `python bench.py --gpus N` without a torchrun environment: it re-executes itself under torchrun)")
    if backend_world is not None and backend_world != world:
        raise SystemExit(f"bench.py: process group has {backend_world} ranks, WORLD_SIZE says {world}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--config", default="MAP3DBN512")
    ap.add_argument("--res", default="512x512", help="output HxW; rays are 3/16 of it per axis (96 for 512)")
    ap.add_argument("--render", default="", help="rays HxW (default: 3/16 of --res per axis, the MAP3DBN512 ratio)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--mode", default="generator", choices=["generator", "dstep", "trainstep"],
                    help="generator: the headline forward benchmark; dstep: BASELINE config 4's discriminator step; "
                         "trainstep: config 4's whole iteration (D step + G step)")
    ap.add_argument("--amp", default="none", choices=["none", "fp16", "bf16"],
                    help="trainstep: autocast type of the library GEMMs / convolutions (the reference's AMP mode is fp16; bf16 "
                         "is there for measurement only: too coarse for the sine layers, and MIOpen's bf16 convolutions are slow)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-telemetry", action="store_true", help="do not sample clocks / power around the timed region")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle-subset self-check of the timed workload")
    ap.add_argument("--check-items", type=int, default=0, help="self-check only the first N items of the batch (0: all; development)")
    a = ap.parse_args()
    respawn_under_torchrun(a, sys.argv[1:])

    rank, world, local, dist_on = dist_env()
    check_world(a, world)
    # MIOpen benchmarks every convolution configuration at first use (minutes for the discriminator's fwd / bwd / double-bwd
    # shapes; its immediate-mode fallback lands on naive kernels: 24 s per D step).  The search results of a previous run
    # are reused (tools/miopen_db, written by MIOpen itself); must be set before the first convolution.
    db = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "miopen_db")
    if os.path.isdir(db):
        os.environ.setdefault("MIOPEN_USER_DB_PATH", db)
    torch.cuda.set_device(local)
    if dist_on:
        init_distributed(local)                                       # RCCL over xGMI
        check_world(a, world, torch.distributed.get_world_size())
    dev = torch.device("cuda", local)
    if a.mode in ("dstep", "trainstep"):
        (discriminator_step_bench if a.mode == "dstep" else train_step_bench)(a, rank, world, dist_on, dev)
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    StageTimer = importlib.import_module("3dhumangan_amd._stages").StageTimer

    H, W = [int(v) for v in a.res.split("x")]
    render = tuple(int(v) for v in a.render.split("x")) if a.render else (H * 3 // 16, W * 3 // 16)
    G, cfg = build_generator(a.config, (H, W), render, a.samples, dev)
    z, cond, jitter = make_inputs(cfg, a.batch, dev, seed=1234 + rank)
    G.stage_timer = StageTimer()
    tel = None
    if rank == 0 and not a.no_telemetry:
        tel = importlib.import_module("3dhumangan_amd._telemetry").Telemetry(local)
    step_ms = []
    dt = timed_steps(G, cfg, z, cond, jitter, a.steps, a.warmup, dist_on, telemetry=tel, step_ms=step_ms)
    torch.cuda.synchronize()
    # drop warm-up samples: keep the last `steps` events of each stage
    stage_ms = {}
    for k, ev in G.stage_timer.events.items():
        ev = ev[-a.steps:]
        stage_ms[k] = (sum(x.elapsed_time(y) for x, y in ev) / len(ev), len(ev))
    G.stage_timer = None
    n_images = a.batch * world * a.steps
    value = n_images / dt

    # every collective of the run is behind us (the MAX over ranks of the elapsed time): ALL ranks leave the process group here,
    # together -- rank 0 then spends a minute on single-rank legs (roofline of A6, the oracle check) and must not find itself
    # tearing down a communicator whose peers exited long ago
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        dist_on = False
    if rank != 0:
        return

    kernels = kernel_rooflines(G, cfg, a.batch, stage_ms)
    kernels["h3d_ray_integrate"] = ray_integrate_roofline(cfg, a.batch)
    dominant = max((k for k in kernels if "frac" in kernels[k] and k != "h3d_ray_integrate"),
                   key=lambda k: kernels[k]["ms"])
    # roofline.achieved = SURVEY 8(d)'s ALGORITHMIC flops of the kernel (what the reference computes per pixel / sample) / its HIP-event
    # time; the flops the kernel EXECUTES after the exact algebraic folding (fewer) are reported next to it as *_executed
    kd = kernels[dominant]
    roof = {k: kd[k] for k in ("bound", "peak", "unit", "engine", "mfma_issue_factor", "mfma_pipe_util", "ms")}
    roof["achieved"] = kd.get("reference_formulation_TFLOPs", kd["achieved"])
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["achieved_executed"], roof["frac_executed"] = kd["achieved"], kd["frac"]
    roof["kernel"] = dominant
    # HBM traffic per launch from the PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE, separate passes,
    # corrected as MI355X_MICROARCH.md prescribes); only valid for the workload it was measured on.
    traffic, traffic_file, busy, traffic_date = load_traffic(f"{a.config}_{H}x{W}_b{a.batch}_s{a.samples}")
    roof["traffic"] = traffic.get(dominant)
    roof["traffic_source"] = (f"{traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a builder run "
                              "(tools/profile_round.sh), NOT measured in this run") if traffic else None
    roof["traffic_source_short"] = (f"{traffic_file} ({traffic_date or 'undated'}): PMC passes of a builder run, NOT this run") if traffic else None
    # matrix-pipe busy fraction of the dominant kernel from the SQ counter pass of the same builder run (SQ_VALU_MFMA_BUSY_CYCLES over
    # the SIMD-cycles of the launch), or None; `mfma_pipe_util` (detail record only) stays the model: issue factor x achieved / peak
    roof["mfma_busy_pmc"] = busy.get(dominant)
    kernels["h3d_ray_integrate"]["traffic"] = traffic.get("h3d_ray_integrate")
    for k, v in traffic.items():
        if k in kernels:
            kernels[k]["traffic"] = v

    extra = {}
    if not a.no_extra and world == 1 and (H, W) == (512, 512):
        def side_run(config, gen_hw, render_hw, samples, batch, steps, engines=None):
            G2, cfg2 = build_generator(config, gen_hw, render_hw, samples, dev)
            if engines:
                G2.neural_field.precision = engines[0]
                G2.synthesis_plan(dev).engine = engines[1]
            z2, cond2, jit2 = make_inputs(cfg2, batch, dev)
            dt2 = timed_steps(G2, cfg2, z2, cond2, jit2, steps, 3, False)
            eng = (G2.neural_field.precision, G2.synthesis_plan(dev).engine)
            del G2
            torch.cuda.empty_cache()
            return dict(images_per_s=batch * steps / dt2, ms_per_step=dt2 / steps * 1e3, batch=batch, engines=eng)

        n2 = max(3, a.steps // 2)
        extra["native_512x256_images_per_s"] = side_run(a.config, (512, 256), (96, 48), a.samples, a.batch, n2)["images_per_s"]
        extra["cfg2_MAP3DBN_256x256_64x64rays_s32"] = side_run("MAP3DBN", (256, 256), (64, 64), 32, 8, n2)
        extra["cfg3L_MAP3DBN512L_512x512_96x96rays_s64"] = side_run("MAP3DBN512L", (512, 512), (96, 96), 64, a.batch, 3)
        extra["cfg5_MAP3DBN512_1024x1024_192x192rays_s128"] = side_run("MAP3DBN512", (1024, 1024), (192, 192), 128, 4, 3)
        extra["headline_workload_on_strict_fp32_mfma_engines"] = side_run(a.config, (H, W), render, a.samples, a.batch, 2,
                                                                        engines=("f32", "f32"))
        extra["headline_workload_on_x3_engines"] = side_run(a.config, (H, W), render, a.samples, a.batch, n2,
                                                          engines=("f16x3", "bf16x3"))
        extra["op_rooflines"] = op_rooflines()
        # BASELINE config 4's per-GPU share (one adversarial iteration: D step with R1 + G step, batch 4, 512x256, 96x48 rays x
        # 32), fp32, MIOpen's search results from tools/miopen_db: the same code path as `--mode trainstep`
        try:
            torch.cuda.reset_peak_memory_stats()
            t4 = train_step_bench(argparse.Namespace(config=a.config, batch=4, steps=3, warmup=2, amp="none"), 0, 1, False, dev, emit=False)
            extra["cfg4_trainstep_b4"] = dict(images_per_s=t4["value"], ms_per_iteration=t4["ms_per_step"],
                                              discriminator_step_ms=t4["stage_ms"]["discriminator_step"],
                                              generator_step_ms=t4["stage_ms"]["generator_step"], peak_memory_GB=t4["peak_memory_GB"],
                                              batch=4, steps=3, warmup=2, dtype=t4["dtype"], workload=t4["config"]["workload"])
            torch.cuda.empty_cache()
            # the same iteration in the reference's AMP mode (float16 autocast + one GradScaler)
            torch.cuda.reset_peak_memory_stats()
            t4h = train_step_bench(argparse.Namespace(config=a.config, batch=4, steps=3, warmup=6, amp="fp16"), 0, 1, False, dev, emit=False)
            extra["cfg4_trainstep_b4_amp_fp16"] = dict(images_per_s=t4h["value"], ms_per_iteration=t4h["ms_per_step"],
                                                       discriminator_step_ms=t4h["stage_ms"]["discriminator_step"],
                                                       generator_step_ms=t4h["stage_ms"]["generator_step"],
                                                       peak_memory_GB=t4h["peak_memory_GB"], batch=4, steps=3, warmup=6)
            torch.cuda.empty_cache()
        except Exception as e:                                  # noqa: BLE001 -- a side workload must not lose the headline
            extra.setdefault("cfg4_trainstep_b4", dict(error=repr(e)[:300]))
            extra["cfg4_trainstep_error"] = repr(e)[:300]

    out = {
        "metric": "generator images/sec at 512^2", "value": value, "unit": "images/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": f"f32 in/out, f32 accumulate; contractions field {G.neural_field.precision} / synthesis "
                 f"{G.synthesis_plan(dev).engine} (x2 = f16 product + block-scaled fp6 cross terms, x3 = three f16/bf16 products)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE cfg 3: {a.config} generator-only forward, {H}x{W} out, {render[0]}x{render[1]} rays x "
                               f"{a.samples} samples, hidden {cfg['hidden_dim']}, batch {a.batch}/GPU, {cfg['map3d_mode']}, random-init "
                               "weights, procedural SMPL-like pose" if (a.config, H, W, a.samples, a.batch) == ("MAP3DBN512", 512, 512, 64, 16)
                               else f"{a.config} generator-only forward, {H}x{W} out, {render[0]}x{render[1]} rays x {a.samples} samples, "
                                    f"hidden {cfg['hidden_dim']}, batch {a.batch}/GPU, {cfg['map3d_mode']}",
                   "global_batch": a.batch * world, "parallelism": f"batch-sharded replicas x{world} (no collective)"},
        "roofline": roof,
        "roofline_hbm_kernel": dict(kernel="h3d_ray_integrate", **{k: kernels["h3d_ray_integrate"][k] for k in
                                    ("bound", "achieved", "peak", "unit", "frac", "traffic")}),
        "kernels": kernels,
        "stage_ms": {k: round(v[0], 4) for k, v in stage_ms.items()},
        "step_ms": step_time_summary(step_ms),
        "telemetry": tel.report() if tel is not None else None,
        "extra": extra,
    }
    # energy per image of THIS rank's GPU over the timed region: median sampled socket power x elapsed time / its images
    try:
        pw = out["telemetry"]["timed"]["socket_power_W"]["median"]
        out["telemetry"]["timed"]["joules_per_image"] = pw * dt / (a.batch * a.steps)
    except (KeyError, TypeError):
        pass
    # every item of the timed batch against the oracle on a >= 5 % patch of its pixels (round 5: 0.12 %)
    out["checked"] = None if a.no_check else self_check(G, cfg, z, cond, jitter, list(range(a.check_items or a.batch)))
    if world == 1 and not a.no_cpu:
        sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
        out["cpu_baseline"] = cpu_baseline(cfg, sd)
        if not a.no_extra:
            out["extra"]["cpu_baseline_cfg1"] = cpu_baseline_cfg1()
            out["extra"]["cpu_baseline_cfg2_b8"] = cpu_baseline_cfg2()
    else:
        out["cpu_baseline"] = None
    write_detail(out)
    sys.stdout.flush()
    print(compact_line(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
