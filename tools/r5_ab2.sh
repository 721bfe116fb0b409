#!/bin/bash
# usage: tools/r5_ab2.sh <tag> "<libs>" [extra bench args]   -- like r5_ab.sh, one repetition, guard off, 2 items checked
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; libs=$2; shift 2
for lib in $libs; do
  name=$(basename $lib .so)
  H3D_SYNTH_GUARD=0 H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 600 python bench.py --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().split("\n")[-1])
    print("$name", d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("checked"))
except Exception as e:
    print("$name failed", e)
PY
done
