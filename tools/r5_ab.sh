#!/bin/bash
# Round-5 development call: same-lease A/B of bench.py between library builds (H3D_LIB), optional quick parity tests first.
# usage: tools/r5_ab.sh <tag> "<lib names relative to 3dhumangan_amd/csrc, space separated>" ["<pytest args>"]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r5ab}; libs=${2:-"libh3d_r4base.so libh3d.so"}; tests=${3:-}
if [ -n "$tests" ]; then
  timeout 900 python -m pytest $tests -x -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
  echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
  tail -4 gpurun_out/${tag}_tests.log
fi
for rep in 1 2; do
for lib in $libs; do
  name=$(basename $lib .so)_$rep
  H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 600 python bench.py --no-cpu --no-extra --no-check --steps 20 --warmup 5 > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  cp bench_detail.json gpurun_out/${tag}_${name}_detail.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().split("\n")[-1])
    print("$name", d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("extra",{}).get("joules_per_image"))
except Exception as e:
    print("$name failed", e)
PY
done
done
