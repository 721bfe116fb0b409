#!/bin/bash
# Round-5 closing call on the GPU box: same-lease A/B of the bounded operand split of the x2 field (library built with
# -DH3D_X2_SPLIT_GENERAL vs the default), the full GPU suite, the driver's bench protocol, the rocprofv3 kernel table of the same
# command, and the full-image x2 error study on four seeds with the ToRGB head tiles.  usage: bash tools/r5_final.sh
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r5f
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for lib in libh3d_gsplit.so libh3d.so; do
  name=$(basename $lib .so)_$rep
  H3D_LIB=$REPO/3dhumangan_amd/csrc/$lib timeout 300 python bench.py --no-cpu --no-extra --check-items 2 --steps 20 --warmup 5 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab_$name.json").read().strip().split("\n")[-1])
    print("$name", d["value"], d["ms_per_step"], d.get("stage_ms"), d["checked"]["max_rel_err"], d["checked"]["max_rel_err_render"])
except Exception as e:
    print("$name failed", e)
PY
done; done
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log
tail -3 $OUT/gpu_suite.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json
tail -c 600 $OUT/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --no-check > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats -name '*.db' | head -1) $OUT/kernel_stats.csv
find $OUT -name '*.db' -delete
rm -rf $OUT/stats
head -8 $OUT/kernel_stats.csv | cut -c1-160
timeout 300 python tools/x2_fullimage_error.py 1234,1,2,3 > $OUT/x2_fullimage_error_heads.txt 2> $OUT/x2_err.err
tail -1 $OUT/x2_fullimage_error_heads.txt
