"""configs mirror: values pinned to a dump of the reference's config dicts (tests/golden/configs.json)."""
import importlib
import json
import os
import types

from conftest import GOLDEN

configs = importlib.import_module("3dhumangan_amd.configs")


def _norm(cfg):
    out = {}
    for k, v in cfg.items():
        key = "step:%d" % k if isinstance(k, int) else k
        if isinstance(v, type):
            v = v.__name__
        out[key] = json.loads(json.dumps(v, default=list))
    return out


def test_config_dicts_match_reference_dump():
    ref = json.load(open(os.path.join(GOLDEN, "configs.json")))
    for name, want in ref["configs"].items():
        got = _norm(getattr(configs, name))
        assert set(got) == set(want), (name, set(got) ^ set(want))
        for k in want:
            assert got[k] == want[k], (name, k, got[k], want[k])


def test_extract_metadata_and_get_config():
    ref = json.load(open(os.path.join(GOLDEN, "configs.json")))
    for key, want in ref["extract_metadata"].items():
        name, step = key.split("@")
        got = _norm(configs.extract_metadata(getattr(configs, name), int(step)))
        assert got == want, key
    opt = types.SimpleNamespace(config="MAP3DBN512L", tune="", variant=0)
    cfg = configs.get_config(opt)
    assert cfg["neural_field_cls"].__name__ == "COORDCONCATSIREN" and cfg["hidden_dim"] == 420 and cfg["legacy_mode"]
    for key, (nxt, last) in ref["upsample_steps"].items():
        name, step = key.split("@")
        got = configs.next_upsample_step(getattr(configs, name), int(step))
        assert (None if got == float("Inf") else got) == nxt, key
        assert configs.last_upsample_step(getattr(configs, name), int(step)) == last, key
