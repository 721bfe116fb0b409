"""Development aid: cycle trace (s_memtime) of one workgroup of the x3 synthesis kernel.
Needs a library built with -DH3D_EXPERIMENT_TRACE (tools/build_variant.sh) and H3D_LIB pointing at it."""
import importlib
import os
import sys

import torch

os.environ["H3D_SYNTH_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
configs = importlib.import_module("3dhumangan_amd.configs")
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
cfg = {k: v for k, v in configs.MAP3DBN512.items() if isinstance(k, str)}
cfg.update(gen_height=512, gen_width=512, dataset_length=2)
cfg["neural_field_cls"] = impl.COORDCONCATSIREN
G = gens.Map3DGenerator(**cfg).to("cuda").eval()
G.set_device("cuda")
fmap = torch.randn(16, 96 * 96, 256, device="cuda")
st = torch.randn(16, 1, 256, device="cuda")
for _ in range(3):
    G._synthesize(fmap, st, (96, 96))
torch.cuda.synchronize()
plan = G.synthesis_plan(fmap.device)
tr = plan.build_x3(plan.engine == "f16x2")["trace"].cpu().tolist()
ev = [(t >> 8, t & 255) for t in tr if t]
t0 = ev[0][0]
prev = t0
for t, tag in ev:
    print(f"{tag} {t - t0:9d} +{t - prev}")
    prev = t
