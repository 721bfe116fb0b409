#!/bin/bash
# Same-lease A/B of bench.py between library builds: bash tools/r5_libs.sh <tag> "<lib names in 3dhumangan_amd/csrc>"  (rep 1 checks 2 items)
cd "$(dirname "$0")/.."
tag=${1:-libs}; libs=${2:-"libh3d.so"}
OUT=$PWD/gpurun_out/$tag
mkdir -p $OUT
for rep in 1 2; do for lib in $libs; do
  name=$(basename $lib .so)_$rep
  chk="--no-check"; [ $rep = 1 ] && chk="--check-items 2"
  H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 300 python bench.py --no-cpu --no-extra $chk --steps 20 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().split("\n")[-1])
    c=d.get("checked") or {}
    print("$name", d["value"], d["ms_per_step"], d["stage_ms"]["render_fused"], d["stage_ms"]["synthesis"], c.get("max_rel_err"), c.get("max_rel_err_render"), c.get("x2_fell_back"))
except Exception as e:
    print("$name failed", e)
PY
done; done
