"""The ONE stdout line of bench.py is what the driver parses (round 4's 25.6 KB line came back `parsed: null`): it must be a
single line, strict JSON (no NaN / Infinity), under 4 KB, carry the contract keys with `roofline` and `cpu_baseline`, and hold no
array longer than 16.  Also: `--gpus N` without a torchrun environment re-executes the script under torch.distributed.run, and
a world size that disagrees with `--gpus` fails loudly."""
import argparse
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _recorded(big=True):
    """A full `out` record as bench.main() assembles it, with everything that made round 4's line 25.6 KB (per-kernel table,
    telemetry series, 36-row convolution table, per-step times, per-item errors) and some non-finite floats thrown in."""
    kernels = {f"h3d_kernel_{i}": dict(bound="mfma", achieved=500.0 + i, peak=2500.0, unit="TFLOP/s", frac=0.2, ms=1.0 + i,
                                       engine="x" * 200, note="y" * 400) for i in range(12 if big else 2)}
    return {
        "metric": "generator images/sec at 512^2", "value": 363.26312345678, "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 44.045234567, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 in/out, f32 accumulate; contractions field f16x2 / synthesis f16x2 (x2 = f16 product + block-scaled fp6 "
                 "cross terms, x3 = three f16/bf16 products)",
        "data": "synthetic",
        "config": {"workload": "MAP3DBN512 generator-only forward, 512x512 output, 96x96 rays, 64 samples/ray, hidden 256, batch "
                               "16/GPU, map3d_mode=mixed, random-init weights, procedural SMPL-like pose",
                   "global_batch": 16, "parallelism": "batch-sharded replicas x1 (no collective)"},
        "roofline": {"kernel": "h3d_synthesis", "bound": "mfma", "achieved": 1016.9, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.4068,
                     "traffic": 8283289600.0, "ms": 24.368, "achieved_executed": 543.0, "frac_executed": 0.2172, "mfma_pipe_util": 0.3258,
                     "engine": "z" * 300, "traffic_source": "w" * 300},
        "roofline_hbm_kernel": {"kernel": "h3d_ray_integrate", "bound": "hbm", "achieved": 6127.2, "peak": 8000.0, "unit": "GB/s",
                                "frac": 0.7659, "traffic": 10800230400.0},
        "kernels": kernels,
        "stage_ms": {"mapping": 0.266, "ray_setup": 0.062, "geo_features": 1.933, "render_fused": 16.738, "synthesis_tables": 0.603,
                     "synthesis": 24.368},
        "step_ms": {"n": 20, "every": [44.0 + 0.01 * i for i in range(40)]},
        "telemetry": {"available": True, "series": [[i, 2100, 1290.0, 70] for i in range(40)],
                      "timed": {"socket_power_W": {"median": 1290.0}, "joules_per_image": float("nan")}},
        "extra": {"native_512x256_images_per_s": 712.48,
                  "cfg2_MAP3DBN_256x256_64x64rays_s32": {"images_per_s": 580.7, "ms_per_step": 13.7, "batch": 8, "engines": ["f16x2t", "f16x2t"]},
                  "cfg3L_MAP3DBN512L_512x512_96x96rays_s64": {"images_per_s": 114.5},
                  "cfg5_MAP3DBN512_1024x1024_192x192rays_s128": {"images_per_s": float("inf")},
                  "headline_workload_on_x3_engines": {"images_per_s": 287.4},
                  "op_rooflines": {"conv_x3_by_shape": [dict(kernel="h3d_conv_x3", shape="B4 512x256 128->128 k3", ms=1.0, frac=0.1)] * 36},
                  "cfg4_trainstep_b4": {"ms_per_iteration": 203.4, "workload": "v" * 300},
                  "cfg4_trainstep_b4_amp_fp16": {"ms_per_iteration": 142.0},
                  "cpu_baseline_cfg1": {"value": 0.0902}, "cpu_baseline_cfg2_b8": {"value": 0.1062}},
        "checked": {"max_rel_err": 6.894e-4, "max_rel_err_render": 1.4e-6, "max_rel_err_image_norm": 4.1e-4, "tolerance": 1e-3, "ok": True,
                    "batch_items": list(range(16)), "per_item_max_rel_err": [1e-4] * 16, "rays_excluded_as_ill_conditioned_in_the_oracle": 0,
                    "pixels_per_item": 13800, "rays_per_item": 532, "pixel_fraction": 0.0526, "x2_fallback_items": [9],
                    "x2_monitor": {"tolerance": 3.5e-4, "max_sampled_err": 4.1e-4, "tiles_per_image": 32}},
        "cpu_baseline": {"value": 0.01256, "unit": "images/s", "cores": 64, "kind": "port", "cpu": "AMD EPYC", "runs": [19.9, 20.1],
                         "sample": "1 image at 1/2 linear size (256x256 px, 48x48 rays x 64 samples, same widths): 1 warm-up + 2 timed "
                                   "runs, median 19.9 s -> 80 s per full-size image (work is linear in rays and pixels); pure-PyTorch "
                                   "CPU oracle, brute-force nearest-vertex search" + " padding" * 100},
    }


def _no_long_arrays(o, limit=16):
    if isinstance(o, dict):
        return all(_no_long_arrays(v, limit) for v in o.values())
    if isinstance(o, list):
        return len(o) <= limit and all(_no_long_arrays(v, limit) for v in o)
    return True


def _strict_loads(text):
    def bad(c):
        raise ValueError(f"non-finite constant {c} in the line")
    return json.loads(text, parse_constant=bad)


def test_line_is_one_short_strict_json_line_with_the_contract_keys():
    out = _recorded()
    assert len(json.dumps(out)) > 12000                        # the record itself is round-4 sized
    text = bench.compact_line(out)
    assert len(text.encode()) < 4096 and "\n" not in text and "\r" not in text
    line = _strict_loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(363.263, abs=1e-3) and line["ms_per_step"] == pytest.approx(44.0452, abs=1e-4)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["higher_is_better"] is True
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["traffic"] == 8283289600.0
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port" and len(c["sample"]) <= 260
    assert line["checked"]["ok"] is True and line["checked"]["tolerance"] == 1e-3 and line["checked"]["items"] == 16
    # round 6: both norms, the norm's name, the checked fraction and the per-item fallbacks are in the contract line
    assert line["checked"]["max_rel_err_image_norm"] == 4.1e-4 and "oracle" in line["checked"]["norm"]
    assert line["checked"]["pixel_fraction_per_item"] >= 0.05 and line["checked"]["x2_fallback_items"] == [9]
    assert line["checked"]["x2_monitor_tol"] == 3.5e-4
    assert _no_long_arrays(line) and len(line["extra"]) <= 10
    assert "kernels" not in line and "telemetry" not in line and "step_ms" not in line
    # the non-finite side numbers were dropped, not printed as NaN / Infinity
    assert "cfg5_1024sq_s128_b4_images_per_s" not in line["extra"] and "joules_per_image" not in line["extra"]


def test_line_survives_absurdly_long_strings():
    out = _recorded()
    out["config"]["workload"] = "w" * 3000
    text = bench.compact_line(out)
    assert len(text) < 4096
    line = _strict_loads(text)
    assert all(k in line for k in CONTRACT)                     # optional parts go first, never the contract keys


def test_line_without_optional_legs():
    out = _recorded(big=False)
    out.update(checked=None, cpu_baseline=None, telemetry=None, extra={})
    line = _strict_loads(bench.compact_line(out))
    assert line["cpu_baseline"] is None and line["checked"] is None and line["extra"] == {}


def test_detail_file_is_strict_json(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    written = bench.write_detail(_recorded())
    assert len(written) == 2
    for f in written:
        d = _strict_loads(open(f).read())
        assert len(d["kernels"]) == 12 and d["telemetry"]["timed"]["joules_per_image"] is None


def test_gpus_n_respawns_under_torchrun(monkeypatch):
    calls = []
    monkeypatch.setattr(os, "execve", lambda exe, cmd, env: calls.append((exe, cmd, env)))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    bench.respawn_under_torchrun(argparse.Namespace(gpus=1), ["--gpus", "1"])
    assert calls == []                                          # one GPU: plain process
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit, match="--gpus 8 but this node shows 4"):
        bench.respawn_under_torchrun(argparse.Namespace(gpus=8), ["--gpus", "8"])       # fewer GPUs than asked for: say so, do not spawn
    assert calls == []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    bench.respawn_under_torchrun(argparse.Namespace(gpus=8), ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    (exe, cmd, env), = calls
    assert exe == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.abspath(bench.__file__), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "8")                       # already under torchrun: never again
    bench.respawn_under_torchrun(argparse.Namespace(gpus=8), ["--gpus", "8"])
    assert calls == []


def test_world_size_must_match_gpus():
    bench.check_world(argparse.Namespace(gpus=4), 4, 4)
    with pytest.raises(SystemExit, match="--gpus 8 but WORLD_SIZE is 1"):
        bench.check_world(argparse.Namespace(gpus=8), 1)
    with pytest.raises(SystemExit, match="process group has 2 ranks"):
        bench.check_world(argparse.Namespace(gpus=4), 4, 2)


def test_ill_conditioned_rays_looks_at_the_oracle_densities_only():
    import torch
    sigma = torch.randn(1, 5, 8) * 3.0
    sigma[0, 0, 0] = 10.0                                       # max |sigma| = 10 -> threshold 1e-2
    sigma[0, :, -1] = torch.tensor([5.0, 9e-3, -9e-3, 1.1e-2, -4.0])
    assert bench.ill_conditioned_rays(sigma).tolist() == [[False, True, True, False, False]]


def test_check_cells_cover_five_percent_and_the_brightest_pixels():
    """Round 6: bench.self_check compares every item on a contiguous patch of >= 5 % of its pixels plus the cells that hold each
    channel's largest |value| -- the oracle's maximum over the checked pixels is then the image's scale (a dim patch alone would
    inflate the relative error, a bright one deflate it)."""
    import torch
    sys.path.insert(0, os.path.join(bench.ROOT, "oracle"))
    import h3d_oracle as O
    cfg = dict(gen_height=512, gen_width=512, render_height=96, render_width=96)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 512, 512, generator=g)
    img[1, 500, 3] = 40.0                                              # a bright pixel near a corner
    bright = bench.brightest_cells(img, (96, 96))
    assert len(bright) == 3
    for item in (0, 7):
        cells, K = bench.check_cells(cfg, item, 0.05, 5, bright)
        pix = O.pixels_of_cells(cells, (512, 512), (96, 96))
        assert K == 22 and len(pix) >= 0.05 * 512 * 512
        for c in range(3):
            assert int(img[c].abs().argmax()) in set(pix.tolist())       # the channel's maximum is among the checked pixels
    a, _ = bench.check_cells(cfg, 0, 0.05, 5)
    b, _ = bench.check_cells(cfg, 1, 0.05, 5)
    assert a != b                                                       # the patch moves from item to item
    # small geometries: the patch never exceeds the render grid
    small = dict(gen_height=16, gen_width=8, render_height=8, render_width=4)
    cells, K = bench.check_cells(small, 0, 0.05, 5)
    assert all(0 <= cy < 8 and 0 <= cx < 4 for cy, cx in cells) and K >= 1
