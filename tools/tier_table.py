"""Error vs throughput of the precision tiers on BASELINE configs 3 and 5 (one MI355X): every engine combination runs the
same batch; error = per-channel max-norm relative deviation of the image / render from the CPU oracle on a pixel / ray subset
(oracle/h3d_oracle.py: generator_forward_subset).  Writes profiles/<tag>_precision_tiers.json.

    python tools/tier_table.py r3"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import h3d_oracle as O                                    # noqa: E402  (checker only; nothing timed goes through it)
from conftest import rel_err_channels                      # noqa: E402
import bench                                               # noqa: E402

COMBOS = [("f16x2", "f16x2"), ("f16x2", "bf16x3"), ("f16x2t", "f16x2t"), ("f16x3", "bf16x3"), ("f16x3t", "bf16x3t"), ("f16x3", "f16w2t"), ("f16x1t", "bf16x3"), ("f16x1t", "f16w2t"),
          ("f16x1t", "f16x1t"), ("f32", "f32")]
WORK = {"cfg3_512sq_b16_s64": ("MAP3DBN512", (512, 512), (96, 96), 64, 16),
        "cfg5_1024sq_b4_s128": ("MAP3DBN512", (1024, 1024), (192, 192), 128, 4)}


def main(tag):
    dev = torch.device("cuda")
    out = {}
    for name, (cfgn, hw, rhw, S, B) in WORK.items():
        G, cfg = bench.build_generator(cfgn, hw, rhw, S, dev)
        z, cond, jit = bench.make_inputs(cfg, B, dev)
        sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
        ocfg = {k: v for k, v in cfg.items() if k != "neural_field_cls"}
        g = torch.Generator().manual_seed(3)
        cells = [(0, 0), (rhw[0] - 1, rhw[1] - 1)] + list(zip(torch.randint(0, rhw[0], (14,), generator=g).tolist(),
                                                              torch.randint(0, rhw[1], (14,), generator=g).tolist()))
        pix = O.pixels_of_cells(cells, hw, rhw)
        refs = []
        for i in (0, B - 1):
            ci = {k: v[i:i + 1].cpu() for k, v in cond.items()}
            refs.append((i, O.generator_forward_subset(sd, ocfg, z[i:i + 1].cpu(), ci, jit[i:i + 1].cpu(), pix)))
        rows = []
        for fe, se in COMBOS:
            G.neural_field.precision = fe
            G.synthesis_plan(dev).engine = se
            for _ in range(2):
                o = G.forward(z, cond, jitter=jit, **cfg)
            torch.cuda.synchronize()
            n = 3 if fe == "f32" else 8
            t0 = time.perf_counter()
            for _ in range(n):
                G.forward(z, cond, jitter=jit, **cfg)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            e_img = e_ren = 0.0
            for i, ref in refs:
                e_img = max(e_img, rel_err_channels(o["rgbs"][i:i + 1].cpu().flatten(2)[:, :, pix], ref["rgbs"]))
                e_ren = max(e_ren, rel_err_channels(o["rgbs_render"][i:i + 1].cpu().flatten(2)[:, :, ref["ray_subset"]], ref["rgbs_render"]))
            rows.append(dict(field=fe, synthesis=se, ms_per_step=ms, images_per_s=B / ms * 1e3, err_image=e_img, err_render=e_ren))
            print(name, rows[-1], flush=True)
        out[name] = rows
        del G
        torch.cuda.empty_cache()
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_precision_tiers.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r2")
