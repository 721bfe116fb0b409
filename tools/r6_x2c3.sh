#!/bin/bash
# Round 6: x2 k-step schedule with the f16 instructions first (record conversions in their shadow) -- parity of the wide x2 tier,
# cfg 3L / cfg 2, per-k-step and per-phase cycle traces at width 384.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_generator.py tests/test_gpu_x2_guard.py -x -q -m gpu -k "x2t or wide" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
  timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra --check-items 2 --steps 5 --warmup 2 > $OUT/L_$rep.json 2> $OUT/L_$rep.err
  timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 > $OUT/2_$rep.json 2> $OUT/2_$rep.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_$rep.json" % k).read().strip().split("\n")[-1])
        c=d.get("checked") or {}
        print(k, "$rep", d["value"], d["ms_per_step"], d.get("stage_ms"), c.get("max_rel_err"), c.get("max_rel_err_render"), c.get("ok"))
    except Exception as e:
        print(k, "$rep failed", e, open("$OUT/%s_$rep.err" % k).read()[-600:])
PY
done 2>&1 | tee $OUT/wide_summary.txt
H3D_LIB=$C/libh3d_tracefine.so timeout 300 python tools/synth_x3t_trace.py MAP3DBN > $OUT/trace_synth_384_x2t_fine.txt 2>&1
H3D_LIB=$C/libh3d_trace.so timeout 300 python tools/synth_x3t_trace.py MAP3DBN > $OUT/trace_synth_384_x2t.txt 2>&1
H3D_FIELD_X3T_WG_PER_CU=0 H3D_LIB=$C/libh3d_trace.so H3D_TRACE_FILE=$OUT/trace_field_384_x2t.txt timeout 300 python tools/field_trace.py 384 > $OUT/trace_384.log 2>&1
sed -n 40,75p $OUT/trace_synth_384_x2t_fine.txt
