#!/bin/bash
# Closing check of round 5 after the per-file scheduler flag: the field / render / ring tests, smoke, and the driver's bench protocol.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r5c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_field.py tests/test_gpu_fused_geo.py tests/test_gpu_precision_tiers.py tests/test_gpu_ring_stress.py "tests/test_gpu_baseline_workloads.py::test_cfg3_bench_workload_b16_512sq" tests/test_gpu_hierarchical.py -x -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"], d["checked"], d["roofline"]["frac"], len(open("$OUT/bench.json").read()))
PY
