#!/bin/bash
# AMP tier: parity tests, then config-4 AMP iteration under (weight planes, dense-layer engine) settings, same lease
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_amp.py tests/test_gpu_conv.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5e_tests.log; tail -5 gpurun_out/r5e_tests.log
for cfg in "2 library" "1 library" "1 x3"; do
  set -- $cfg
  H3D_AMP_WEIGHT_PLANES=$1 H3D_AMP_LINEAR=$2 timeout 600 python bench.py --mode trainstep --amp fp16 --batch 4 --steps 5 --warmup 6 > gpurun_out/r5e_amp_p$1_$2.json 2> gpurun_out/r5e_amp_p$1_$2.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5e_amp_p$1_$2.json").read().strip().split("\n")[-1])
    print("planes=$1 linear=$2", round(d["ms_per_step"],2), d.get("stage_ms"), round(d.get("peak_memory_GB",0),2))
except Exception as e:
    print("planes=$1 linear=$2 failed", e)
PY
done
