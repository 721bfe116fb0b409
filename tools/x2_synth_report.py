"""GPU: bench-geometry generator forward with the synthesis engine on bf16x3 vs f16x2 (field engine fixed): image error against
the CPU oracle subset (bench.py self_check) and stage times.  usage: python tools/x2_synth_report.py [batch]"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
StageTimer = importlib.import_module("3dhumangan_amd._stages").StageTimer
out = {}
for field, synth in (("f16x3", "bf16x3"), ("f16x2", "bf16x3"), ("f16x2", "f16x2")):
    G, cfg = bench.build_generator("MAP3DBN512", (512, 512), (96, 96), 64, dev)
    G.neural_field.precision = field
    G.synthesis_plan(dev).engine = synth
    z, cond, jitter = bench.make_inputs(cfg, B, dev)
    G.stage_timer = StageTimer()
    dt = bench.timed_steps(G, cfg, z, cond, jitter, 10, 3, False)
    torch.cuda.synchronize()
    st = {k: round(sum(a.elapsed_time(b) for a, b in v[-10:]) / 10, 3) for k, v in G.stage_timer.events.items()}
    G.stage_timer = None
    chk = bench.self_check(G, cfg, z, cond, jitter, sorted({0, B - 1}), n_cells=48)
    out[f"{field}/{synth}"] = dict(images_per_s=B * 10 / dt, stage_ms=st, max_rel_err=chk["max_rel_err"], max_rel_err_render=chk["max_rel_err_render"],
                                   pixels=chk["pixels"])
    print(field, synth, json.dumps(out[f"{field}/{synth}"]), flush=True)
