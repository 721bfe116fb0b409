import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# MIOpen (the discriminator's convolutions): reuse the shipped search results for config 4's shapes instead of benchmarking
# every configuration at first use (minutes); must be set before the first convolution runs
_MIOPEN_DB = os.path.join(ROOT, "tools", "miopen_db")
if os.path.isdir(_MIOPEN_DB):
    os.environ.setdefault("MIOPEN_USER_DB_PATH", _MIOPEN_DB)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / driver GPU tier)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """-> nested dict of torch tensors from tests/golden/<name>.npz ('a/b' keys nest)."""
    raw = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in raw.files:
        node = out
        parts = k.split("/")
        # state-dict keys contain dots only, so '/' is safe as the nesting separator
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        v = raw[k]
        if parts[-1].endswith("_json"):
            node[parts[-1][:-5]] = json.loads(bytes(v).decode())
        else:
            node[parts[-1]] = torch.from_numpy(np.array(v))
    return out


@pytest.fixture(scope="session")
def h3d():
    return importlib.import_module("3dhumangan_amd")


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).detach())


def rel_err_channels(a, b, dim=1):
    """max over channels (dimension `dim`) of max|a-b| / max|b| within the channel: a low-magnitude channel cannot hide
    behind a large one, unlike the single max-norm of rel_err."""
    a, b = a.double().transpose(0, dim).flatten(1), b.double().transpose(0, dim).flatten(1)
    return float(((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)).max())


def rel_err_rms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).square().mean().sqrt() / b.square().mean().sqrt().clamp_min(1e-30))


def grad_errors(got, ref, zero_below=1e-4, skip=()):
    """Per-parameter max-norm relative error of gradient dicts.  Gradients that are mathematically zero (a conv bias in front
    of a batch-statistics BatchNorm) hold rounding noise (<= 2e-5 in the reference's own autograd): for those only the
    magnitude is checked.  -> (worst relative error, name of the worst)"""
    worst, worst_name = 0.0, None
    for k, r in ref.items():
        if k in skip:
            continue
        g = got[k]
        assert g is not None, f"no gradient for {k}"
        g, r = g.detach().cpu(), r.detach().cpu()
        assert g.shape == r.shape, k
        if float(r.abs().max()) < zero_below:
            assert float(g.abs().max()) < zero_below, (k, float(g.abs().max()))
            continue
        e = rel_err(g, r)
        if e > worst:
            worst, worst_name = e, k
    return worst, worst_name
