#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats of the same command, separate PMC passes for HBM traffic and for the SQ
# counters, the same for the wide (x3t) workload.  usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r2
set -u
R=${1:-r5}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
KERN='x3_kernel|x3t_kernel|geo_features|mesh_sort|ray_integrate|conv_x3|wgrad|synthesis_check'
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err        # the driver's protocol
cp bench_detail.json $OUT/bench_detail.json                                              # the full record behind the compact line
python bench.py --steps 200 --warmup 5 --no-extra --no-cpu --no-check > $OUT/bench_200steps.json 2>> $OUT/bench.err
cp bench_detail.json $OUT/bench_200steps_detail.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --no-check > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p -- python $REPO/bench.py --no-cpu --no-extra --no-check --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o p -- python $REPO/bench.py --no-cpu --no-extra --no-check --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_sq1.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o p -- python $REPO/bench.py --no-cpu --no-extra --no-check --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_sq2.err
# the wide workload (MAP3DBN512L, hidden 420: the x3t engines)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_wide -o k -- python $REPO/bench.py --config MAP3DBN512L --no-cpu --no-extra --no-check --steps 5 > $OUT/bench_wide_under_rocprof.json 2> $OUT/stats_wide.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/pmc_wide -o p -- python $REPO/bench.py --config MAP3DBN512L --no-cpu --no-extra --no-check --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_wide.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats -name '*.db' | head -1) $OUT/kernel_stats.csv
python tools/rocprof_summary.py $(find $OUT/stats_wide -name '*.db' | head -1) $OUT/wide_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE sq1 sq2 wide; do
  python tools/pmc_dump.py $(find $OUT/pmc_$c -name '*.db' | head -1) "$KERN" > $OUT/pmc_$c.txt
done
python tools/traffic_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt MAP3DBN512_512x512_b16_s64 $OUT/hbm_traffic.json
# BASELINE config 4: one adversarial iteration per step (MIOpen's search results come from tools/miopen_db)
python bench.py --mode trainstep --batch 4 --steps 5 --warmup 2 > $OUT/trainstep_1gpu.json 2> $OUT/trainstep.err
python tools/train_profile.py 4 g > $OUT/trainstep_gstep_kernels.txt 2>> $OUT/trainstep.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_train -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 2 > $OUT/trainstep_bench_under_rocprof.json 2>> $OUT/trainstep.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats_train -name '*.db' | head -1) $OUT/trainstep_kernel_stats.csv
# the same iteration in the reference's AMP mode: bench line + kernel table
python bench.py --mode trainstep --batch 4 --steps 5 --warmup 6 --amp fp16 > $OUT/trainstep_1gpu_amp_fp16.json 2>> $OUT/trainstep.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_train_amp -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 6 --amp fp16 > /dev/null 2>> $OUT/trainstep.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats_train_amp -name '*.db' | head -1) $OUT/trainstep_amp_fp16_kernel_stats.csv
find $OUT -name '*.db' -delete
tail -c 400 $OUT/bench.json
