"""Optional per-stage HIP-event timing of the generator forward (used by bench.py; zero cost when disabled)."""
import contextlib

import torch


class StageTimer:
    """Collects (start, end) event pairs per named stage on the current stream; durations are read after a sync."""

    def __init__(self):
        self.events = {}

    @contextlib.contextmanager
    def stage(self, name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            self.events.setdefault(name, []).append((a, b))

    def summary_ms(self):
        """-> {stage: (mean ms per call, calls)}; call after torch.cuda.synchronize()."""
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self.events.items()}

    def reset(self):
        self.events = {}


@contextlib.contextmanager
def stage(owner, name):
    t = getattr(owner, "stage_timer", None)
    if t is None:
        yield
    else:
        with t.stage(name):
            yield
