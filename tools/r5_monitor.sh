#!/bin/bash
# monitor tests + same-lease A/B of the bench with the x2 monitor on / off
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x2_monitor.py tests/test_gpu_x2_guard.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5d_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5d_tests.log; tail -6 gpurun_out/r5d_tests.log
for rep in 1 2; do for mon in 0 1; do
  H3D_SYNTH_MONITOR=$mon timeout 600 python bench.py --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 > gpurun_out/r5d_mon${mon}_$rep.json 2> gpurun_out/r5d_mon${mon}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5d_mon${mon}_$rep.json").read().strip().split("\n")[-1])
    print("monitor=$mon", d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("checked"))
except Exception as e:
    print("monitor=$mon failed", e)
PY
done; done
