#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do for heads in 0 1; do
  H3D_SYNTH_HEADS=$heads timeout 600 python bench.py --no-cpu --no-extra --check-items 4 --steps 20 --warmup 5 > gpurun_out/r5k_heads${heads}_$rep.json 2> gpurun_out/r5k_heads${heads}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5k_heads${heads}_$rep.json").read().strip().split("\n")[-1])
    print("heads=$heads", d["value"], d["ms_per_step"], d.get("stage_ms"), d["checked"])
except Exception as e:
    print("heads=$heads failed", e)
PY
done; done
timeout 500 python -m pytest tests/test_gpu_x2_guard.py tests/test_gpu_x2_monitor.py "tests/test_gpu_baseline_workloads.py::test_cfg3_bench_workload_b16_512sq" "tests/test_gpu_baseline_workloads.py::test_cfg3_native_aspect_b16" -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
