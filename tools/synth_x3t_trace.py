"""Development aid: cycle trace (s_memtime) of workgroup (1000, 3) of the LDS-resident synthesis kernel (synthesis_x3t.hip) at
MAP3DBN's width (384; the trace buffer does not fit the LDS next to a 448-wide tile).  Needs a library built with
-DH3D_EXPERIMENT_TRACE (tools/build_x3t_variant.sh trace "-DH3D_EXPERIMENT_TRACE") and H3D_LIB pointing at it: the trace
overwrites the first bytes of the output image."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
configs = importlib.import_module("3dhumangan_amd.configs")
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
name = sys.argv[1] if len(sys.argv) > 1 else "MAP3DBN"
cfg = {k: v for k, v in getattr(configs, name).items() if isinstance(k, str)}
cfg.update(gen_height=256, gen_width=256, dataset_length=2)
cfg["neural_field_cls"] = impl.COORDCONCATSIREN
G = gens.Map3DGenerator(**cfg).to("cuda").eval()
G.set_device("cuda")
F = cfg["feature_dim"]
fmap = torch.randn(8, 64 * 64, F, device="cuda")
st = torch.randn(8, 1, F, device="cuda")
for _ in range(3):
    rgb = G._synthesize(fmap, st, (64, 64))
torch.cuda.synchronize()
print("engine", G.synthesis_plan(fmap.device).engine)
tr = rgb.flatten()[:1000].contiguous().view(torch.int64).cpu().tolist()
ev = [(t >> 8, t & 255) for t in tr if t > 0 and (t & 255) < 64]
if ev:
    t0 = prev = ev[0][0]
    for t, tag in ev:
        if t < prev or t - t0 > 10_000_000:
            break
        print(f"{tag} {t - t0:9d} +{t - prev}")
        prev = t
