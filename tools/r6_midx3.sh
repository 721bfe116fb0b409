#!/bin/bash
# Round 6: the residual stream's base block on three bf16 products inside the x2 synthesis kernel (H3D_SYNTH_MID_X3=1, default) vs all-x2 (=0): parity
# suites, same-lease bench A/B, full-image error against the x3 engines with the monitor's sampled errors.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_x2_monitor.py tests/test_gpu_x2_guard.py tests/test_gpu_generator.py tests/test_gpu_baseline_workloads.py tests/test_gpu_ring_stress.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do for mid in 0 1; do
  H3D_SYNTH_MID_X3=$mid timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu --check-items 16 > $OUT/bench_mid${mid}_$rep.json 2> $OUT/bench_mid${mid}_$rep.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_mid${mid}_$rep.json").read().strip().split("\n")[-1]); c=d["checked"]
print("mid_x3=$mid rep $rep", d["value"], d["ms_per_step"], d["stage_ms"]["synthesis"], "err", c["max_rel_err"], "img", c["max_rel_err_image_norm"], "monitor", c["x2_monitor_err"], "redo", c["x2_fallback_items"], "ok", c["ok"])
PY
done; done 2>&1 | tee $OUT/midx3_ab.txt
for mid in 0 1; do H3D_SYNTH_MID_X3=$mid timeout 900 python tools/x2_fullimage_error.py 1234,1,2,3,7,8 > $OUT/fullimage_mid$mid.txt 2>&1; tail -8 $OUT/fullimage_mid$mid.txt | cut -c1-400; done
