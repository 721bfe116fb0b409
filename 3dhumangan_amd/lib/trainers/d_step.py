"""One discriminator optimisation step: the D half of BASELINE config 4 (reference: PhaseTrainer.train_discriminator /
_train_discriminator, lib/trainers/phase_trainer.py:297-318, 344-430), batch-sharded over the GPUs of a node.

The fake images come from the HIP generator under no_grad (the reference also generates them under no_grad, :357-379); the
discriminator runs through torch autograd.  Multi-GPU: every rank holds a batch shard; the ONLY data-path collectives are
  * the all-gather of the R1 statistics (parallel.r1_allgather, RCCL over xGMI) so that every rank applies the same
    global penalty, and
  * the all-reduce of the discriminator gradients, bucketed and launched from autograd hooks while backward is still
    running (parallel.GradReducer), the exchange DDP does implicitly in the reference.
"""
import torch

from ... import parallel
from . import losses


def _stepped_since_update(scaler, optimizer):
    """True when `optimizer` was unscaled / stepped through `scaler` and scaler.update() has not run since.  GradScaler keeps
    that per optimizer in ``_per_optimizer_states[id(optimizer)]["stage"]`` (READY / UNSCALED / STEPPED; update() resets it) --
    private state, so its shape is CHECKED: if a torch release moves it, this raises instead of silently answering False (the
    second D-alone step would then die inside unscale_ with a message about something else);
    tests/test_scaler_state_cpu.py pins it."""
    states = getattr(scaler, "_per_optimizer_states", None)
    if states is None or not hasattr(states, "get"):
        raise RuntimeError("GradScaler no longer exposes _per_optimizer_states: discriminator_step cannot tell whether the "
                           "scale must be updated before this step; call scaler.update() yourself after every step")
    st = states.get(id(optimizer))
    if st is None:
        return False
    stage = st.get("stage") if hasattr(st, "get") else None
    if stage is None or not hasattr(stage, "name"):
        raise RuntimeError("GradScaler's per-optimizer state has no 'stage': see _stepped_since_update")
    return stage.name != "READY"


def discriminator_step(D, optimizer, real_images, fake_images, gt_segments, meta, do_r1=True, r1_mode="reference",
                       distributed=False, grad_clip=None, amp_dtype=None, scaler=None, update_scaler=False):
    """-> dict of detached scalars.  meta: gan_lambda, segmentation_lambda, r1_lambda, label_dim (config keys).

    ``r1_mode``: "reference" (default, drop-in) is the reference's penalty -- the channel norms of sample 0 of the shard
    (losses.py); "per_sample" is the textbook statistic, opt-in: with the shipped ``r1_lambda`` it is about C times stronger.
    Either statistic is all-gathered over the ranks (parallel.r1_allgather) and the penalty is the mean over ALL gathered
    values: for "reference" that equals the mean over ranks of the per-rank penalties, which is what the reference gets from
    DDP's gradient averaging; for "per_sample" it is the global-batch mean.

    ``amp_dtype`` / ``scaler``: the reference's AMP mode (autocast around the discriminator forwards; the R1 gradient is taken
    of the SCALED prediction sum and unscaled afterwards, phase_trainer.py:270-283).  The reference shares ONE GradScaler
    between both steps and updates it once per iteration, in train_generator (phase_trainer.py:335-338); its D step only
    calls scaler.step -- so ``update_scaler`` is off by default here.  (Changed in round 3: the defaults used to be
    r1_mode="per_sample" and an update per D step.)  A caller that runs D steps ALONE, or several per G step, never reaches
    that update: the step notices that this optimizer was already stepped since the scaler's last update (GradScaler would
    raise on the second unscale_) and updates the scale itself first."""
    if scaler is not None and _stepped_since_update(scaler, optimizer):
        scaler.update()
    amp = dict(device_type="cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None)
    gan_lambda, seg_lambda = meta["gan_lambda"], meta["segmentation_lambda"]
    optimizer.zero_grad(set_to_none=True)
    real = real_images.detach()
    if do_r1:
        real = real.requires_grad_(True)                 # the input gradient is only needed by the penalty
    with torch.autocast(**amp):
        out_real = D(real, None, 1.0)
        out_fake = D(fake_images.detach(), None, 1.0)
    out_real = {k: v.float() for k, v in out_real.items()}
    out_fake = {k: v.float() for k, v in out_fake.items()}
    gan = losses.logistic_d_loss(out_real["prediction"], out_fake["prediction"], gan_lambda) if gan_lambda > 0 else \
        (out_real["prediction"].sum() + out_fake["prediction"].sum()) * 0
    penalty, r1_scale = real.new_zeros(()), 1.0
    if do_r1:
        # the loss scale as a device tensor (scaler.scale(1)): no host synchronisation, unlike scaler.get_scale()
        scale = scaler.scale(real.new_ones(())) if scaler is not None else None
        grad = losses.r1_gradient(real, out_real, gan_lambda, scale=scale)
        stat = losses.r1_statistic(grad, r1_mode)
        if distributed:
            # all ranks' statistics; only this rank's slice carries a graph.  The "reference" statistic is C values on every
            # rank: one collective, no length exchange, no host synchronisation
            stat = parallel.r1_allgather(stat, equal=(r1_mode == "reference") or bool(meta.get("equal_shards", False)))
            # the mean over the gathered values already divides this rank's share by the world size; the gradient all-reduce
            # below AVERAGES over the ranks (right for the per-shard means gan / seg), so the R1 term is pre-multiplied by it
            r1_scale = float(torch.distributed.get_world_size())
        penalty = 0.5 * meta["r1_lambda"] * stat.mean()
        penalty = torch.where(torch.isnan(penalty), torch.zeros_like(penalty), penalty)   # the reference's NaN guard (:291)
    seg = real.new_zeros(())
    if seg_lambda > 0 and out_real["segments"].shape[1] > 0:
        s_real, acc, _ = losses.segmentation_loss(out_real["segments"], gt_segments, meta["label_dim"], meta.get("segmentation_weights"))
        s_gen, _, _ = losses.segmentation_loss(out_fake["segments"], torch.zeros_like(gt_segments), meta["label_dim"],
                                               meta.get("segmentation_weights"))
        seg = (s_real + s_gen) * seg_lambda
    loss = gan + 4 * penalty + seg                       # lazy regularisation factor of the reference (:392)
    total = gan + 4 * r1_scale * penalty + seg
    # the gradient all-reduce runs inside backward, bucket by bucket (parallel.GradReducer: the exchange DDP overlaps for the
    # reference, lib/trainers/base_trainer.py:102-104)
    reducer = parallel.reducer_of(D) if distributed else None
    if reducer is not None:
        reducer.prepare()
    (scaler.scale(total) if scaler is not None else total).backward()
    if reducer is not None:
        reducer.finish()
    if scaler is not None:
        scaler.unscale_(optimizer)
    if grad_clip is not None:
        torch.nn.utils.clip_grad_norm_(D.parameters(), grad_clip)
    if scaler is not None:
        scaler.step(optimizer)
        if update_scaler:
            scaler.update()
    else:
        optimizer.step()
    return dict(loss=loss.detach(), gan=gan.detach(), r1=penalty.detach(), segmentation=seg.detach())
