#!/bin/bash
# Round 6, first lease: what bounds the wide (x3t) engines?  (a) same-lease bench of cfg 3L / cfg 2 on the shipped library and on
# the ALIAS build (every hidden matrix reads the same 0.8 MB: an L2-resident weight stream, wrong results) -- the difference is
# what the misses of the 5.6 / 15 MB streams cost; (b) cycle trace of one workgroup of the fused field kernel at width 420.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r6a
mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
for rep in 1 2; do for lib in libh3d.so libh3d_alias.so; do
  name=$(basename $lib .so)_$rep
  H3D_LIB=$C/$lib timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra --no-check --steps 5 --warmup 2 > $OUT/L_$name.json 2> $OUT/L_$name.err
  H3D_LIB=$C/$lib timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --no-check --steps 20 --warmup 5 > $OUT/2_$name.json 2> $OUT/2_$name.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_$name.json" % k).read().strip().split("\n")[-1])
        print(k, "$name", d["value"], d["ms_per_step"], d.get("stage_ms"))
    except Exception as e:
        print(k, "$name failed", e)
PY
done; done 2>&1 | tee $OUT/summary.txt
for w in 420 384; do
  H3D_LIB=$C/libh3d_trace.so H3D_TRACE_FILE=$OUT/trace_field_${w}_x2t.txt timeout 300 python tools/field_trace.py $w > $OUT/trace_$w.log 2>&1
  H3D_FIELD_PRECISION=f16x3t H3D_LIB=$C/libh3d_trace.so H3D_TRACE_FILE=$OUT/trace_field_${w}_x3t.txt timeout 300 python tools/field_trace.py $w >> $OUT/trace_$w.log 2>&1
done
head -40 $OUT/trace_field_420_x2t.txt
