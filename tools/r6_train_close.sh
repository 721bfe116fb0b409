#!/bin/bash
# config-4 closing numbers of round 6: bench lines (fp32, AMP, one RCCL rank under torchrun) + rocprofv3 kernel tables of both modes.
# usage (GPU box, repo root): bash tools/r6_train_close.sh <tag>
set -u
R=${1:-r6t}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
python bench.py --mode trainstep --batch 4 --steps 8 --warmup 2 > $OUT/trainstep_1gpu.json 2> $OUT/trainstep.err
python bench.py --mode trainstep --batch 4 --steps 8 --warmup 6 --amp fp16 > $OUT/trainstep_1gpu_amp_fp16.json 2>> $OUT/trainstep.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --mode trainstep --gpus 1 --batch 4 --steps 5 --warmup 2 > $OUT/trainstep_torchrun_1rank_rccl.json 2>> $OUT/trainstep.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_train -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 2 > $OUT/trainstep_bench_under_rocprof.json 2>> $OUT/trainstep.err
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_train_amp -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 6 --amp fp16 > /dev/null 2>> $OUT/trainstep.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats_train -name '*.db' | head -1) $OUT/trainstep_kernel_stats.csv
python tools/rocprof_summary.py $(find $OUT/stats_train_amp -name '*.db' | head -1) $OUT/trainstep_amp_fp16_kernel_stats.csv
python tools/train_profile.py 4 both none shapes > $OUT/trainstep_fp32_ops_by_shape.txt 2>> $OUT/trainstep.err
find $OUT -name '*.db' -delete
for f in trainstep_1gpu trainstep_1gpu_amp_fp16 trainstep_torchrun_1rank_rccl; do tail -1 $OUT/$f.json | cut -c1-60; python -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'], d.get('peak_memory_GB'))"; done
