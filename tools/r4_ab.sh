#!/bin/bash
# Round-4 development call: targeted parity tests of the reworked register engines, then a same-lease A/B of the bench
# (this tree vs the round-3 tree: recreate it with `git worktree add _r3_baseline 943f8f6 && (cd _r3_baseline && python -c "import __graft_entry__ as g; g.build()")`).  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r4a}
timeout 900 python -m pytest tests/test_gpu_x2_guard.py tests/test_gpu_ring_stress.py tests/test_gpu_field.py tests/test_gpu_conv.py \
  "tests/test_gpu_generator.py" -x -q -m gpu -s -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
for arm in new old new2; do
  if [ $arm = old ]; then dir=_r3_baseline; else dir=.; fi
  [ -d $dir ] || continue
  (cd $dir && timeout 600 python bench.py --no-cpu --no-extra --steps 10 --warmup 3) > gpurun_out/${tag}_bench_$arm.json 2> gpurun_out/${tag}_bench_$arm.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_bench_$arm.json").read().strip().split("\n")[-1])
    print("$arm", d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d.get("stage_ms",{}).items()}, d.get("checked"))
except Exception as e:
    print("$arm failed", e)
PY
done
