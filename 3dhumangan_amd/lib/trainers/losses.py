"""Losses of the discriminator step (reference: lib/trainers/phase_trainer.py:203-294, 344-430), as plain functions.

R1: the reference's `_calculate_r1_regularization` (:259-294) does NOT compute the textbook per-sample penalty.  Its line
`grad_real = [p * inv_scale for p in grad_real][0]` iterates over the BATCH dimension of the gradient and keeps sample 0, and
the following `view(size(0), -1).pow(2).sum(1).mean()` then averages the squared norms of that one sample's CHANNELS:
    penalty_ref = 0.5 * r1_lambda * mean_c sum_{h,w} grad[0, c, h, w]^2
`r1_statistic(..., mode="reference")` reproduces that bit for bit (drop-in parity, the default of discriminator_step);
mode="per_sample" is the textbook ||grad_x D||^2 per sample (opt-in: about C times stronger at the same r1_lambda).  In
the multi-GPU step either statistic is all-gathered across the ranks (parallel.r1_allgather).
"""
import contextlib
import os

import torch
import torch.nn.functional as F

from ..components.ops import conv as conv_ops

R1_INPUT_GRADS_ONLY = os.environ.get("H3D_R1_INPUT_GRADS_ONLY", "1") != "0"      # A/B switch (round 6)


def logistic_d_loss(pred_real, pred_gen, gan_lambda=1.0):
    """softplus(D(fake)) + softplus(-D(real)), means over every prediction pixel (phase_trainer.py:388-389)."""
    return gan_lambda * (F.softplus(pred_gen).mean() + F.softplus(-pred_real).mean())


def r1_gradient(d_input_real, d_output_real, gan_lambda=1.0, scale=None):
    """d sum(D(x).prediction) / dx with the graph kept (double backward); softmax(segments) when the GAN head is off.
    ``scale``: the GradScaler's loss scale under fp16 AMP -- the target is scaled before the backward pass (so that half-
    precision gradients inside the discriminator do not underflow) and the result unscaled, as phase_trainer.py:270-283."""
    if gan_lambda > 0:
        target = d_output_real["prediction"].sum()
    else:
        target = torch.softmax(d_output_real["segments"], dim=1).sum()
    # only the image's gradient is asked for: the native convolutions skip their weight / bias gradients in this pass (the engine
    # would drop them unread; ops/conv.py: input_grads_only)
    with conv_ops.input_grads_only() if R1_INPUT_GRADS_ONLY else contextlib.nullcontext():
        if scale is None or (not torch.is_tensor(scale) and scale == 1.0):
            return torch.autograd.grad(outputs=target, inputs=d_input_real, create_graph=True)[0]
        grad = torch.autograd.grad(outputs=target * scale, inputs=d_input_real, create_graph=True)[0]
    return grad * (1.0 / scale)                     # `scale` may be a device scalar: no host round trip


def r1_statistic(grad, mode="per_sample"):
    """grad [B,C,H,W] -> the squared-norm statistics whose mean is the penalty (x 0.5 * r1_lambda): [B] per-sample norms, or
    the reference's [C] channel norms of sample 0."""
    if mode == "per_sample":
        return grad.flatten(1).pow(2).sum(dim=1)
    if mode == "reference":
        g0 = grad[0]
        return g0.reshape(g0.shape[0], -1).pow(2).sum(dim=1)
    raise ValueError(f"unknown R1 mode {mode!r}")


def r1_penalty(grad, r1_lambda, mode="per_sample"):
    return 0.5 * r1_lambda * r1_statistic(grad, mode).mean()


def segmentation_loss(segments, gt_segments, label_dim, prior_weights=None):
    """Class-balanced cross entropy of the discriminator's segmentation head (phase_trainer.py:204-256, mode
    "cross_entropy_balanced") -> (loss, accuracy over the foreground classes, mean foreground probability)."""
    B, _, H, W = segments.shape
    if gt_segments.shape[1] != H or gt_segments.shape[2] != W:
        gt_segments = F.interpolate(gt_segments.unsqueeze(1).float(), (H, W), mode="nearest").squeeze(1).long()
    pw = torch.ones(label_dim, dtype=segments.dtype, device=segments.device) if prior_weights is None else \
        torch.as_tensor(prior_weights, dtype=segments.dtype, device=segments.device)
    pw = pw / pw.mean()
    if torch.any(gt_segments > 0):
        one_hot = F.one_hot(gt_segments, num_classes=label_dim).permute(0, 3, 1, 2)
        occ = one_hot.sum(dim=(0, 2, 3))
        occ[0] = 0
        n_present = torch.count_nonzero(occ)
        coef = torch.reciprocal(occ.to(segments.dtype)) * one_hot.numel() / (n_present * one_hot.shape[1])
        coef[0] = 0
        coef[torch.isinf(coef)] = 0
        loss = (F.cross_entropy(segments, gt_segments, reduction="none") * (coef * pw)[gt_segments]).mean()
    else:
        loss = F.cross_entropy(segments, gt_segments)
    real_prob = (1 - torch.softmax(segments, dim=1)[:, 0]).mean()
    accuracy = ((torch.argmax(segments[:, 1:], dim=1) + 1) == gt_segments).float().mean()
    return loss, accuracy, real_prob
