#!/bin/bash
# PMC passes of the tree that closes round 5 (bench workload only): HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes) and
# the two SQ counter sets, each in its own rocprofv3 run without trace options.  usage: bash tools/r5_pmc.sh
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r5p
mkdir -p $OUT
KERN='x3_kernel|geo_features|mesh_sort|ray_integrate|synthesis_check'
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu --no-extra --no-check --steps 2 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p -- $B > /dev/null 2> $OUT/pmc_$c.err
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o p -- $B > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o p -- $B > /dev/null 2> $OUT/pmc_sq2.err
cd $REPO
for c in FETCH_SIZE WRITE_SIZE sq1 sq2; do
  python tools/pmc_dump.py $(find $OUT/pmc_$c -name '*.db' | head -1) "$KERN" > $OUT/pmc_$c.txt
done
python tools/traffic_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt MAP3DBN512_512x512_b16_s64 $OUT/hbm_traffic.json > /dev/null
find $OUT -name '*.db' -delete
grep -A9 "synthesis_x3_kernel<8, 4, false, true, true>" $OUT/pmc_sq1.txt | head -10
grep -A9 "synthesis_x3_kernel<8, 4, false, true, true>" $OUT/pmc_sq2.txt | head -10
