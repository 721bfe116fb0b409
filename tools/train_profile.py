"""Steady-state kernel breakdown of the config-4 training iteration (bench.py --mode trainstep) with torch.profiler: MIOpen's
find-mode kernels of the warm-up steps stay out of the table.  usage: python tools/train_profile.py [batch] [g|d|both] [none|fp16] [shapes]
(`shapes`: also the input shapes of the matrix products / convolutions that reached the library)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = sys.argv[2] if len(sys.argv) > 2 else "both"
amp = sys.argv[3] if len(sys.argv) > 3 else "none"
shapes = len(sys.argv) > 4 and sys.argv[4] == "shapes"
amp_dtype = torch.float16 if amp == "fp16" else None
scaler = torch.amp.GradScaler("cuda") if amp == "fp16" else None
dev = "cuda"
trainers = importlib.import_module("3dhumangan_amd.lib.trainers")
disc = importlib.import_module("3dhumangan_amd.lib.discriminators")
G, cfg = bench.build_generator("MAP3DBN512", (512, 256), (96, 48), 32, dev)
G.train()
z, cond, jitter = bench.make_inputs(cfg, batch, dev)
torch.manual_seed(99)
D = disc.UNetDiscriminator(**{k: v for k, v in cfg.items() if k != "neural_field_cls"}).to(dev)
meta = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
meta.update(gan_lambda=1.0, segmentation_lambda=1.0, r1_lambda=10.0, gen_lr=5e-5, betas=(0.0, 0.9))
opt_d = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.0, 0.9))
opt_g = trainers.make_generator_optimizer(G, meta)
g = torch.Generator().manual_seed(7)
real = torch.randn(batch, 3, 512, 256, generator=g).clamp(-1, 1).to(dev)
gt = torch.randint(0, max(1, cfg.get("label_dim", 1)), (batch, 512, 256), generator=g).to(dev)
fwd = {k: v for k, v in cfg.items() if isinstance(k, str)}


def step():
    if which in ("d", "both"):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp_dtype is not None):
            fake = G(z, cond, jitter=jitter, **fwd)["rgbs"]
        trainers.discriminator_step(D, opt_d, real, fake, gt, meta, do_r1=True, grad_clip=cfg.get("grad_clip", 10.0),
                                    amp_dtype=amp_dtype, scaler=scaler)
    if which in ("g", "both"):
        trainers.generator_step(G, D, opt_g, z, cond, meta, gt_segments=gt, generator_kwargs=dict(jitter=jitter),
                                amp_dtype=amp_dtype, scaler=scaler)


for _ in range(6 if amp == "fp16" else 3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=shapes) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / N / 1e3, e.count // N) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
total = sum(r[1] for r in rows if not r[0].startswith(("aten::", "autograd::", "_", "Optimizer")))
print(f"# steady-state device time per iteration (batch {batch}, part {which}); kernels only sum to {total:.1f} ms")
for k, ms, n in rows[:70]:
    print(f"{ms:9.2f} ms  x{n:<5d} {k[:150]}")
if shapes:
    print("# library matrix products / convolutions by input shape")
    lib_ops = ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::convolution_backward", "aten::miopen_convolution",
               "aten::_convolution")
    srows = [(e.key, str(e.input_shapes), e.device_time_total / N / 1e3, e.count // N)
             for e in prof.key_averages(group_by_input_shape=True) if e.key in lib_ops and e.device_time_total > 0]
    srows.sort(key=lambda r: -r[2])
    for k, shp, ms, n in srows[:40]:
        print(f"{ms:9.2f} ms  x{n:<4d} {k:28s} {shp}")
    print("# type conversions / copies by input shape")
    crow = [(e.key, str(e.input_shapes), e.device_time_total / N / 1e3, e.count // N)
            for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::_to_copy", "aten::copy_", "aten::contiguous", "aten::clone")
            and e.device_time_total > 0]
    crow.sort(key=lambda r: -r[2])
    for k, shp, ms, n in crow[:40]:
        print(f"{ms:9.2f} ms  x{n:<4d} {k:28s} {shp}")
    print("# element-wise / reduction / resampling operators by input shape (self device time)")
    skip = set(lib_ops) | {"aten::_to_copy", "aten::copy_", "aten::contiguous", "aten::clone"}
    erow = [(e.key, str(e.input_shapes)[:110], e.self_device_time_total / N / 1e3, e.count // N)
            for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.key not in skip
            and e.self_device_time_total > 0]
    erow.sort(key=lambda r: -r[2])
    for k, shp, ms, n in erow[:70]:
        print(f"{ms:9.2f} ms  x{n:<4d} {k:34s} {shp}")
