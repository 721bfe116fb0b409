"""One discriminator optimisation step: the D half of BASELINE config 4 (reference: PhaseTrainer.train_discriminator /
_train_discriminator, lib/trainers/phase_trainer.py:297-318, 344-430), batch-sharded over the GPUs of a node.

The fake images come from the HIP generator under no_grad (the reference also generates them under no_grad, :357-379); the
discriminator runs through torch autograd.  Multi-GPU: every rank holds a batch shard; the ONLY data-path collectives are
  * the all-gather of the per-sample R1 statistics (parallel.r1_allgather, RCCL over xGMI) so that every rank applies the
    same global-batch penalty, and
  * the all-reduce of the discriminator gradients (parallel.allreduce_gradients), the exchange DDP does implicitly in the
    reference.
"""
import torch

from ... import parallel
from . import losses


def discriminator_step(D, optimizer, real_images, fake_images, gt_segments, meta, do_r1=True, r1_mode="per_sample",
                       distributed=False, grad_clip=None, amp_dtype=None, scaler=None):
    """-> dict of detached scalars.  meta: gan_lambda, segmentation_lambda, r1_lambda, label_dim (config keys).
    r1_mode "reference" reproduces the reference's penalty on this rank's shard exactly (see losses.py); "per_sample" gathers
    the per-sample squared gradient norms of ALL ranks and penalises their global mean.  ``amp_dtype`` / ``scaler``: the
    reference's AMP mode (autocast around the discriminator forwards; the R1 gradient is taken of the SCALED prediction sum and
    unscaled afterwards, phase_trainer.py:270-283)."""
    amp = dict(device_type="cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None)
    gan_lambda, seg_lambda = meta["gan_lambda"], meta["segmentation_lambda"]
    optimizer.zero_grad(set_to_none=True)
    real = real_images.detach().requires_grad_(True)
    with torch.autocast(**amp):
        out_real = D(real, None, 1.0)
        out_fake = D(fake_images.detach(), None, 1.0)
    out_real = {k: v.float() for k, v in out_real.items()}
    out_fake = {k: v.float() for k, v in out_fake.items()}
    gan = losses.logistic_d_loss(out_real["prediction"], out_fake["prediction"], gan_lambda) if gan_lambda > 0 else \
        (out_real["prediction"].sum() + out_fake["prediction"].sum()) * 0
    penalty, r1_scale = real.new_zeros(()), 1.0
    if do_r1:
        scale = scaler.get_scale() if scaler is not None else 1.0
        grad = losses.r1_gradient(real, out_real, gan_lambda, scale=scale)
        stat = losses.r1_statistic(grad, r1_mode)
        if distributed and r1_mode == "per_sample":
            stat = parallel.r1_allgather(stat)           # [world * b]; only this rank's slice carries a graph
            # the global mean already divides this rank's share by world * b; the gradient all-reduce below AVERAGES over
            # the ranks (right for the per-shard means gan / seg), so the R1 term is pre-multiplied by world
            r1_scale = float(torch.distributed.get_world_size())
        penalty = 0.5 * meta["r1_lambda"] * stat.mean()
        if torch.isnan(penalty):
            penalty = real.new_zeros(())
    seg = real.new_zeros(())
    if seg_lambda > 0 and out_real["segments"].shape[1] > 0:
        s_real, acc, _ = losses.segmentation_loss(out_real["segments"], gt_segments, meta["label_dim"], meta.get("segmentation_weights"))
        s_gen, _, _ = losses.segmentation_loss(out_fake["segments"], torch.zeros_like(gt_segments), meta["label_dim"],
                                               meta.get("segmentation_weights"))
        seg = (s_real + s_gen) * seg_lambda
    loss = gan + 4 * penalty + seg                       # lazy regularisation factor of the reference (:392)
    total = gan + 4 * r1_scale * penalty + seg
    (scaler.scale(total) if scaler is not None else total).backward()
    if distributed:
        parallel.allreduce_gradients(D.parameters(), average=True)
    if scaler is not None:
        scaler.unscale_(optimizer)
    if grad_clip is not None:
        torch.nn.utils.clip_grad_norm_(D.parameters(), grad_clip)
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    return dict(loss=loss.detach(), gan=gan.detach(), r1=penalty.detach(), segmentation=seg.detach())
