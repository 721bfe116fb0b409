#!/bin/bash
# Round 6: last-sample refinement + per-item flags on hardware, bench line, and the same-lease A/B of the weight prefetch in synthesis_x3t.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r6d
mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
timeout 1800 python -m pytest tests/test_gpu_refine.py tests/test_gpu_fused_geo.py tests/test_gpu_x2_monitor.py tests/test_gpu_x2_guard.py tests/test_gpu_baseline_workloads.py tests/test_gpu_generator.py -x -q -m gpu -s > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; tail -3 $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json
for rep in 1 2; do for lib in libh3d_nopre.so libh3d.so; do
  name=$(basename $lib .so)_$rep
  H3D_LIB=$C/$lib timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra --check-items 1 --steps 5 --warmup 2 > $OUT/L_$name.json 2> $OUT/L_$name.err
  H3D_LIB=$C/$lib timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --check-items 1 --steps 20 --warmup 5 > $OUT/2_$name.json 2> $OUT/2_$name.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_$name.json" % k).read().strip().split("\n")[-1])
        c=d.get("checked") or {}
        print(k, "$name", d["value"], d["ms_per_step"], d.get("stage_ms"), c.get("max_rel_err"), c.get("max_rel_err_render"), c.get("ok"))
    except Exception as e:
        print(k, "$name failed", e)
PY
done; done 2>&1 | tee $OUT/wide_summary.txt
