"""Exponential moving average of the generator parameters -- counterpart of the reference's lib/components/ema.py:8-93
(same attribute names, so the `*_ema.pth` files BaseTrainer.save_model pickles restore into this class through
checkpoints.load_reference_pickle).  The shadow list is POSITIONAL over `parameters()` filtered by requires_grad; this
build's Map3DGenerator registers its parameters in the reference's order (tests/golden/param_order.json pins it).
"""
import torch


class ExponentialMovingAverage:

    def __init__(self, parameters, decay, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in parameters if p.requires_grad]
        self.collected_params = []

    def current_decay(self):
        """min(decay, (1 + n) / (10 + n)) with n the update count AFTER this update (warm-up of the reference)."""
        if self.num_updates is None:
            return self.decay
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    @torch.no_grad()
    def update(self, parameters):
        if self.num_updates is not None:
            self.num_updates += 1
        rate = 1.0 - self.current_decay()
        live = [p.detach() for p in parameters if p.requires_grad]
        # same association as the reference, s -= (1-d) * (s - p), as three multi-tensor kernels instead of ~3 per parameter
        diff = torch._foreach_sub(self.shadow_params, live)
        torch._foreach_mul_(diff, rate)
        torch._foreach_sub_(self.shadow_params, diff)

    @torch.no_grad()
    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, (q for q in parameters if q.requires_grad)):
            p.copy_(s)

    def store(self, parameters):
        self.collected_params = [p.detach().clone() for p in parameters if p.requires_grad]

    @torch.no_grad()
    def restore(self, parameters):
        for c, p in zip(self.collected_params, (q for q in parameters if q.requires_grad)):
            p.copy_(c)

    def to(self, device):
        self.shadow_params = [p.to(device) for p in self.shadow_params]
        self.collected_params = [p.to(device) for p in self.collected_params]
        return self
