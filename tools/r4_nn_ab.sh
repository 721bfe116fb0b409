#!/bin/bash
# A/B of the nearest-vertex pruning on one lease: bench line with H3D_NN_PRUNE=0 / 1, alternating.  usage: bash tools/r4_nn_ab.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for p in 0 1 0 1; do
  H3D_NN_PRUNE=$p python bench.py --steps 20 --warmup 5 --no-extra --no-cpu --no-check > gpurun_out/r4_nn_ab_$p.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r4_nn_ab_$p.json"))
print("prune=$p", round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.get("stage_ms", {}).items()})
PY
done
