#!/bin/bash
# Round 6, second lease of the wide-engine study: (a) cycle trace of one workgroup of synthesis_x3t at width 384; (b) same-lease
# bench of cfg 3L / cfg 2 on the shipped library and on the NO_WREC build (the weights' second plane -- the fp6 records of the x2
# tier -- is never loaded: half the weight bytes through the vector-memory path, wrong results): is the GEMM loop bound by the
# 64 B/clk/CU of the L1 path?
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r6b
mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
H3D_LIB=$C/libh3d_trace.so timeout 300 python tools/synth_x3t_trace.py MAP3DBN > $OUT/trace_synth_384_x2t.txt 2>&1
H3D_SYNTH_PRECISION=bf16x3t H3D_LIB=$C/libh3d_trace.so timeout 300 python tools/synth_x3t_trace.py MAP3DBN > $OUT/trace_synth_384_x3t.txt 2>&1
for rep in 1 2; do for lib in libh3d.so libh3d_nowrec.so; do
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
head -150 $OUT/trace_synth_384_x2t.txt
