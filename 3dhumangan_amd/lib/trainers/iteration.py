"""One whole adversarial iteration of BASELINE config 4 = discriminator step + generator step (reference:
PhaseTrainer.train_discriminator / train_generator, lib/trainers/phase_trainer.py:297-341), as `bench.py --mode trainstep` times
it and as the world-2 gloo test runs it: one function, so what is measured on N GPUs is what is tested on N CPU ranks.

Collectives of one iteration on N > 1 ranks (every rank holds a batch shard):
  generator forward (twice: under no_grad for the D step, recorded for the G step)   [2C+1] BatchNorm moment all-reduces per SPADE
  discriminator step                                                                  all-gather of the R1 statistics,
                                                                                      bucketed gradient all-reduce inside backward
  generator step                                                                      BatchNorm backward-moment all-reduces,
                                                                                      bucketed gradient all-reduce inside backward
"""
import torch

from .d_step import discriminator_step
from .g_step import generator_step


def adversarial_iteration(G, D, opt_d, opt_g, z, conditions, real_images, gt_segments, meta, generator_kwargs=None, ema=None,
                          distributed=False, grad_clip=None, amp_dtype=None, scaler=None, r1_mode="reference", do_r1=True,
                          on_phase=None):
    """-> (discriminator-step scalars, generator-step scalars).  ``on_phase(name)`` is called before the D step ("d"), between
    the steps ("g") and at the end ("end"): the bench records its HIP events there."""
    mark = on_phase or (lambda name: None)
    fwd = {k: v for k, v in meta.items() if isinstance(k, str)}
    fwd.update(generator_kwargs or {})
    mark("d")
    with torch.no_grad(), torch.autocast("cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None):
        fake = G(z, conditions, **fwd)["rgbs"].float()
    d = discriminator_step(D, opt_d, real_images, fake, gt_segments, meta, do_r1=do_r1, r1_mode=r1_mode, distributed=distributed,
                           grad_clip=grad_clip, amp_dtype=amp_dtype, scaler=scaler)
    mark("g")
    g = generator_step(G, D, opt_g, z, conditions, meta, gt_segments=gt_segments, ema=ema, distributed=distributed,
                       generator_kwargs=generator_kwargs, amp_dtype=amp_dtype, scaler=scaler)
    mark("end")
    return d, g
