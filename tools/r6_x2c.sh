#!/bin/bash
# Round 6: the compact (x2c) weight format of the wide engines' x2 tier on hardware: parity suites of the wide engines, then cfg 3L / cfg 2.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_field.py tests/test_gpu_generator.py tests/test_gpu_x2_guard.py tests/test_gpu_precision_tiers.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python -m pytest tests/test_gpu_baseline_workloads.py -x -q -m gpu -k "cfg2 or cfg3L or 3L or wide" > $OUT/pytest2.log 2>&1; tail -3 $OUT/pytest2.log
for rep in 1 2; do
  timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra --check-items 2 --steps 5 --warmup 2 > $OUT/L_$rep.json 2> $OUT/L_$rep.err
  timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 > $OUT/2_$rep.json 2> $OUT/2_$rep.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_$rep.json" % k).read().strip().split("\n")[-1])
        c=d.get("checked") or {}
        print(k, "$rep", d["value"], d["ms_per_step"], d.get("stage_ms"), c.get("max_rel_err"), c.get("max_rel_err_render"), c.get("ok"), c.get("rays_excluded"))
    except Exception as e:
        print(k, "$rep failed", e, open("$OUT/%s_$rep.err" % k).read()[-600:])
PY
done 2>&1 | tee $OUT/wide_summary.txt
