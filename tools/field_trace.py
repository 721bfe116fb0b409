"""Development aid: cycle trace of one workgroup of the fused x3 field kernel (library built with
-DH3D_EXPERIMENT_TRACE via tools/build_variant.sh, H3D_LIB pointing at it).  The trace buffer travels in the unused
`out` pointer of the fused entry point."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("3dhumangan_amd._lib")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
dev = "cuda"
B, R, S = 8, 9216, 64
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256        # width: 256 -> field_x3, > 256 -> field_x3t (or H3D_FIELD_PRECISION)
net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=H, hidden_dim=H, geo_feature_dim=31, output_dim=H + 4, feature_dim=H,
                            num_blocks=4).to(dev).eval()
N = R * S
pts = torch.rand(B, N, 3, device=dev) * 2 - 1
geo = torch.rand(B, N, 31, device=dev) * 2 - 1
fr = torch.randn(B, 4 * H, device=dev) * 0.5
ph = torch.randn(B, 4 * H, device=dev)
z = torch.sort(torch.rand(B, R, S, 1, device=dev) + 11, dim=2).values
for _ in range(2):
    net.render(pts, fr, ph, geo, None, z, S, input_scaler=0.7, last_back=True, white_back=True)
torch.cuda.synchronize()
print("trace is written through Args.out: see tools/field_trace.py docstring")
