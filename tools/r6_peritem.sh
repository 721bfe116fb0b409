#!/bin/bash
# Round 6: per-item guard / monitor flags, the >= 5 % oracle check and the full-size CPU baseline on hardware.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r6c
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_x2_monitor.py tests/test_gpu_x2_guard.py tests/test_gpu_baseline_workloads.py -x -q -m gpu -s > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json
