#!/bin/bash
# PMC counters of the weight-gradient kernels on the per-shape table.  usage: bash tools/r4_wgrad_pmc.sh
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_wgrad_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $OUT/p1 -o p -- python $REPO/tools/conv_table.py weight_gradient > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o p -- python $REPO/tools/conv_table.py weight_gradient > /dev/null 2> $OUT/p2.err
cd $REPO
for c in p1 p2; do python tools/pmc_dump.py $(find $OUT/$c -name '*.db' | head -1) "wgrad" > $OUT/$c.txt; done
find $OUT -name '*.db' -delete
cat $OUT/p1.txt $OUT/p2.txt | head -150
