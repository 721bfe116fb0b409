"""One generator optimisation step: the G half of BASELINE config 4 (reference: PhaseTrainer.train_generator /
_train_generator, lib/trainers/phase_trainer.py:321-341, 444-545; optimiser groups :53-71), batch-sharded over the GPUs of a
node.

The generator runs in train mode through its differentiable path (lib/generators/differentiable.py: library GEMMs + HIP
activation / integration kernels with hand-written adjoints, batch-statistics BatchNorm synchronised over the process group),
the discriminator through torch autograd.  Multi-GPU: every rank holds a batch shard; the data-path collectives are the
BatchNorm moment all-reduces inside the forward / backward (2C floats per layer) and the bucketed all-reduce of the generator
gradients (parallel.allreduce_gradients), the exchange DDP does implicitly in the reference.
"""
import math
import os

import torch
import torch.nn.functional as F

from ... import parallel
from . import losses


def generator_param_groups(G, meta):
    """The five Adam groups of the reference (phase_trainer.py:53-71): synthesis side at gen_lr, appearance codes, the two
    mapping networks and the implicit function at their own multiples of it."""
    named = list(G.named_parameters())
    nf_map = {n: p for n, p in named if "neural_field_mapping_network" in n}
    syn_map = {n: p for n, p in named if "synthesis_mapping_network" in n}
    codes = {n: p for n, p in named if "latent_pool" in n}
    nf = {n: p for n, p in named if "neural_field" in n and n not in nf_map}
    taken = {**codes, **nf, **nf_map, **syn_map}
    rest = {n: p for n, p in named if n not in taken}
    lr = meta["gen_lr"]
    return [
        {"params": list(rest.values()), "name": "generator", "lr": lr},
        {"params": list(codes.values()), "name": "appearance_codes", "lr": lr * meta.get("appearance_codes_lr_mul", 1.0)},
        {"params": list(nf_map.values()), "name": "neural_field_mapping", "lr": lr * meta.get("mapping_net_lr_mul", 1.0)},
        {"params": list(syn_map.values()), "name": "synthesis_mapping", "lr": lr},
        {"params": list(nf.values()), "name": "neural_field", "lr": lr * meta.get("neural_field_lr_mul", 1.0)},
    ]


def make_generator_optimizer(G, meta):
    return torch.optim.Adam(generator_param_groups(G, meta), lr=meta["gen_lr"], betas=tuple(float(b) for b in meta["betas"]),
                            weight_decay=meta.get("weight_decay", 0))


def topk_count(meta, d_step_count, batch):
    """phase_trainer.py:487-492: the share of the batch (best-scoring fakes) the GAN loss is taken over."""
    if "topk_interval" in meta and "topk_v" in meta:
        share = max(0.99 ** (d_step_count / meta["topk_interval"]), meta["topk_v"])
    else:
        share = 1.0
    return math.ceil(share * batch)


# The reference leaves the discriminator's parameters trainable during the generator step (phase_trainer.py:326-337): autograd then
# computes every weight gradient of D for a loss that only optimizer_G steps on, and the next discriminator step zeroes them unread
# (:302).  Nothing observable depends on them, so by default they are not computed here (the parameters are frozen for the duration
# of the step: ~1/4 of D's weight-gradient work per iteration); H3D_G_STEP_D_GRADS=1 / d_param_grads=True computes them as the
# reference does.
G_STEP_D_GRADS = os.environ.get("H3D_G_STEP_D_GRADS", "0") == "1"


def generator_step(G, D, optimizer, z, conditions, meta, gt_segments=None, ema=None, distributed=False, d_step_count=0,
                   gen_modal="rgbs", latent_indices=None, generator_kwargs=None, amp_dtype=None, scaler=None,
                   update_scaler=True, d_param_grads=None):
    """-> dict of detached scalars.  meta: the config dict (gan_lambda, segmentation_lambda, label_dim, grad_clip and every
    forward key of the generator).  ``gt_segments`` [B,H,W] int64 (the rasterised body-part labels of the conditions) feeds
    the segmentation term; the unconditional phase of the reference (latent_lambda = perceptual = photometric = 0 in every
    shipped config).

    ``amp_dtype`` (torch.float16 / torch.bfloat16) runs both networks under autocast -- the reference's AMP mode
    (base_trainer.py:50-51, torch.cuda.amp.autocast + GradScaler): the library GEMMs and convolutions run in that type, the
    HIP kernels between them compute in fp32; pass a torch.amp.GradScaler as ``scaler`` for float16."""
    gan_lambda, seg_lambda = meta.get("gan_lambda", 0), meta.get("segmentation_lambda", 0)
    optimizer.zero_grad(set_to_none=True)
    fwd = {k: v for k, v in meta.items() if isinstance(k, str)}
    fwd.update(generator_kwargs or {})
    fwd.update(latent_indices=latent_indices, disable_synthesis=(gen_modal != "rgbs"))
    frozen = [] if (G_STEP_D_GRADS if d_param_grads is None else d_param_grads) else [p for p in D.parameters() if p.requires_grad]
    for p in frozen:
        p.requires_grad_(False)
    try:
        return _generator_step(G, D, optimizer, z, conditions, meta, gt_segments, ema, distributed, d_step_count, gen_modal, fwd,
                               gan_lambda, seg_lambda, amp_dtype, scaler, update_scaler)
    finally:
        for p in frozen:
            p.requires_grad_(True)


def _generator_step(G, D, optimizer, z, conditions, meta, gt_segments, ema, distributed, d_step_count, gen_modal, fwd, gan_lambda,
                    seg_lambda, amp_dtype, scaler, update_scaler):
    with torch.autocast("cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None):
        out = G(z, conditions, **fwd)
        d_out = D(out[gen_modal], conditions, 1.0)
        pred = d_out["prediction"].float()
        k = topk_count(meta, d_step_count, pred.shape[0])
        pred = torch.topk(pred, k, dim=0).values
        gan = gan_lambda * F.softplus(-pred).mean() if gan_lambda > 0 else pred.sum() * 0
        seg = pred.new_zeros(())
        if "segments" in d_out and d_out["segments"].shape[1] > 0:
            if seg_lambda > 0 and gt_segments is not None:
                seg = losses.segmentation_loss(d_out["segments"].float(), gt_segments, meta["label_dim"],
                                               meta.get("segmentation_weights"))[0] * seg_lambda
            else:
                seg = d_out["segments"].sum() * 0
        latent = d_out["latents"].sum() * 0 if "latents" in d_out else 0.0
        loss = gan + seg + latent
    # gradient all-reduce overlapped with backward, rank-invariant set (parallel.GradReducer)
    reducer = parallel.reducer_of(G) if distributed else None
    if reducer is not None:
        reducer.prepare()
    (scaler.scale(loss) if scaler is not None else loss).backward()
    if reducer is not None:
        reducer.finish()
    if scaler is not None:
        scaler.unscale_(optimizer)
    if meta.get("grad_clip") is not None:
        torch.nn.utils.clip_grad_norm_(G.parameters(), meta["grad_clip"])
    if scaler is not None:
        scaler.step(optimizer)
        if update_scaler:
            scaler.update()                # once per iteration, here: the one scaler is shared with the D step (:335-338)
    else:
        optimizer.step()
    if ema is not None:
        ema.update(G.parameters())
    return dict(loss=loss.detach(), gan=gan.detach(), segmentation=seg.detach(), topk=k)
