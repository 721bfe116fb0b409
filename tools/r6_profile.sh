#!/bin/bash
# Round-6 profile round (one lease): the driver's bench line + detail, rocprofv3 kernel stats of the same command, separate PMC
# passes (HBM traffic: FETCH_SIZE, WRITE_SIZE; two SQ counter sets), the wide workload (MAP3DBN512L) the same way, cfg 4's two
# training lines with kernel stats, the one-rank RCCL lines (generator bench and trainstep under torchrun).
# usage (GPU box, repo root): bash tools/r6_profile.sh r6p
set -u
R=${1:-r6p}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
KERN='x3_kernel|x3t_kernel|geo_features|mesh_sort|ray_integrate|conv_x3|wgrad|synthesis_check'
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err        # the driver's protocol
cp bench_detail.json $OUT/bench_detail.json
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu --no-extra --no-check"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- $B --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o p -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_sq1.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o p -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_sq2.err
# the wide workloads (x3t engines): cfg 3L (MAP3DBN512L, hidden 420) and cfg 2 (MAP3DBN, 384)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_wide -o k -- $B --config MAP3DBN512L --steps 5 > $OUT/bench_wide_under_rocprof.json 2> $OUT/stats_wide.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/pmc_wide -o p -- $B --config MAP3DBN512L --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_wide.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats -name '*.db' | head -1) $OUT/kernel_stats.csv
python tools/rocprof_summary.py $(find $OUT/stats_wide -name '*.db' | head -1) $OUT/wide_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE sq1 sq2 wide; do
  python tools/pmc_dump.py $(find $OUT/pmc_$c -name '*.db' | head -1) "$KERN" > $OUT/pmc_$c.txt
done
SCLK=$(python -c "import json; print(json.load(open('$OUT/bench_detail.json'))['telemetry']['timed']['gfxclk_MHz']['median'])" 2>/dev/null || echo 2100)
python tools/traffic_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt MAP3DBN512_512x512_b16_s64 $OUT/hbm_traffic.json $OUT/pmc_sq2.txt $OUT/kernel_stats.csv $SCLK > /dev/null
# BASELINE config 4: one adversarial iteration per step, fp32 and the reference's AMP mode, each with its kernel table
python bench.py --mode trainstep --batch 4 --steps 5 --warmup 2 > $OUT/trainstep_1gpu.json 2> $OUT/trainstep.err
python bench.py --mode trainstep --batch 4 --steps 5 --warmup 6 --amp fp16 > $OUT/trainstep_1gpu_amp_fp16.json 2>> $OUT/trainstep.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_train -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 2 > /dev/null 2>> $OUT/trainstep.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_train_amp -o k -- python $REPO/bench.py --mode trainstep --batch 4 --steps 3 --warmup 6 --amp fp16 > /dev/null 2>> $OUT/trainstep.err
cd $REPO
python tools/rocprof_summary.py $(find $OUT/stats_train -name '*.db' | head -1) $OUT/trainstep_kernel_stats.csv
python tools/rocprof_summary.py $(find $OUT/stats_train_amp -name '*.db' | head -1) $OUT/trainstep_amp_fp16_kernel_stats.csv
# one-rank RCCL lines: the driver's N > 1 command line with one rank (process group over RCCL, barrier, MAX all-reduce of the time)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu --check-items 2 > $OUT/bench_torchrun_1rank_rccl.json 2> $OUT/torchrun.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --mode trainstep --batch 4 --steps 3 --warmup 2 > $OUT/trainstep_torchrun_1rank_rccl.json 2>> $OUT/torchrun.err
find $OUT -name '*.db' -delete
tail -c 600 $OUT/bench.json; echo; tail -c 300 $OUT/bench_torchrun_1rank_rccl.json; echo; tail -c 300 $OUT/trainstep_torchrun_1rank_rccl.json
