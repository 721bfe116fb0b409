"""The D step reads GradScaler's private per-optimizer stage to know whether the scale must be updated first (a caller that runs
D steps alone never reaches the reference's one update per iteration).  That private state is pinned here: if a torch release
renames it, this test fails -- and the helper raises instead of answering False."""
import importlib

import pytest
import torch

d_step = importlib.import_module("3dhumangan_amd.lib.trainers.d_step")


def test_stage_is_visible_and_tracks_step_and_update():
    w = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.SGD([w], lr=0.1)
    scaler = torch.amp.GradScaler("cpu", init_scale=4.0)
    assert d_step._stepped_since_update(scaler, opt) is False              # never stepped
    scaler.scale((w * w).sum()).backward()
    scaler.step(opt)
    assert d_step._stepped_since_update(scaler, opt) is True               # stepped, not yet updated
    with pytest.raises(RuntimeError):
        scaler.unscale_(opt)                                               # what a second D-alone step would run into
    scaler.update()
    assert d_step._stepped_since_update(scaler, opt) is False
    opt.zero_grad()
    scaler.scale((w * w).sum()).backward()
    scaler.unscale_(opt)
    assert d_step._stepped_since_update(scaler, opt) is True               # unscaled counts too


def test_missing_private_state_raises_instead_of_answering_false():
    class Moved:
        pass
    with pytest.raises(RuntimeError, match="_per_optimizer_states"):
        d_step._stepped_since_update(Moved(), object())

    class NoStage:
        _per_optimizer_states = {1: {}}
    opt = type("O", (), {})()
    NoStage._per_optimizer_states = {id(opt): {"found_inf_per_device": {}}}
    with pytest.raises(RuntimeError, match="stage"):
        d_step._stepped_since_update(NoStage(), opt)
