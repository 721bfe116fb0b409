#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_geo.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do for tiled in 0 1; do
  H3D_NN_TILED=$tiled timeout 600 python bench.py --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 > gpurun_out/r5i_tiled${tiled}_$rep.json 2> gpurun_out/r5i_tiled${tiled}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5i_tiled${tiled}_$rep.json").read().strip().split("\n")[-1])
    print("tiled=$tiled", d["value"], d["ms_per_step"], d.get("stage_ms"), d["checked"]["max_rel_err"])
except Exception as e:
    print("tiled=$tiled failed", e)
PY
done; done
