#!/bin/bash
# rocprofv3 kernel stats of the trainstep bench, fp32 vs AMP fp16.  usage (GPU box, repo root): bash tools/r4_amp_prof.sh <tag>
set -u
T=${1:-r4amp}
OUT=$PWD/gpurun_out/$T
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for m in none fp16; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_$m -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 4 --warmup 6 --amp $m > $OUT/train_$m.json 2> $OUT/train_$m.err
  python $REPO/tools/rocprof_summary.py $(find $OUT/stats_$m -name '*.db' | head -1) $OUT/kernels_$m.csv
done
find $OUT -name '*.db' -delete
head -45 $OUT/kernels_fp16.csv
