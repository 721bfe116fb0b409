#!/bin/bash
# Same-lease A/B of the wide (x3t) engines between library builds on cfg 3L (MAP3DBN512L, hidden 420) and cfg 2 (MAP3DBN, 384, 256^2 x 8).
# usage: bash tools/r5_wide.sh "<lib names in 3dhumangan_amd/csrc>"
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r5w
mkdir -p $OUT
for rep in 1 2; do for lib in $1; do
  name=$(basename $lib .so)_$rep
  chk="--no-check"; [ $rep = 1 ] && chk="--check-items 1"
  H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra $chk --steps 5 --warmup 2 > $OUT/L_$name.json 2> $OUT/L_$name.err
  H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra $chk --steps 20 --warmup 5 > $OUT/2_$name.json 2> $OUT/2_$name.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_$name.json" % k).read().strip().split("\n")[-1])
        c=d.get("checked") or {}
        print(k, "$name", d["value"], d["ms_per_step"], d.get("stage_ms"), c.get("max_rel_err"), c.get("max_rel_err_render"))
    except Exception as e:
        print(k, "$name failed", e)
PY
done; done
