"""A deterministic stand-in for the generator / preprocessor, shared by make_golden.py (which drives the
REFERENCE's apps.sample_from_generator.generate_frames with it) and tests/test_app.py (which drives ours).
It records what the harness feeds in and returns an image that is a pure function of it, so the harness logic
(seed -> z, angle schedule, clamp / uint8 conversion) is pinned independently of any RNG inside a real generator."""
import torch


class StubGenerator:
    device = "cpu"

    def __init__(self):
        self.calls = []

    def staged_forward(self, z, conditions, **config):
        H, W = config["gen_height"], config["gen_width"]
        c2w = conditions["cam2world_matrices"]
        self.calls.append((z.clone(), c2w.clone()))
        yy = torch.linspace(-1.5, 1.5, H).view(1, 1, H, 1)
        xx = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W)
        base = torch.stack([z[:, :3].mean(), c2w[:, 0, 3].mean(), c2w[:, 2, 2].mean()]).view(1, 3, 1, 1)
        return {"rgbs": base + yy * xx + 0.3 * torch.sin(5 * xx + z[:, 0].view(-1, 1, 1, 1))}


class StubPreprocessor:
    """forward_with_rotation: encodes the three angles into cam2world, returns a non-trivial semantics map."""

    def forward_with_rotation(self, data, h, v, r, **kw):
        data = dict(data)
        B = h.shape[0]
        m = torch.eye(4).repeat(B, 1, 1)
        m[:, 0, 3] = h.reshape(B)
        m[:, 1, 3] = v.reshape(B)
        m[:, 2, 2] = torch.cos(h.reshape(B)) + r.reshape(B)
        data["cam2world_matrices"] = m
        sem = torch.zeros(B, 3, kw["gen_height"], kw["gen_width"])
        sem[:, 0, ::2] = 0.5
        sem[:, 1, :, 1::3] = -2.0
        data["rasterized_semantics"] = sem
        return data
