#!/bin/bash
# Round 6: does a second wave per SIMD hide one wave's VALU / LDS / DMA issue under the other's MFMAs?
# tools/probes/occupancy_probe.hip: the engines' own GEMM loop (gemm_x3_roll: LDS-DMA weight ring, ds_read_b128 fragments, 6 MFMAs per
# section) with NF filler VALU instructions per section, one (OCC=1: 4 waves per CU) or two (OCC=2: 8 waves per CU, <= 256 registers
# each) workgroups per CU.  Same total work in both.  Prints microseconds per k-step of CU time; then the SQ counters of the two
# NF = 36 arms (6 VALU per MFMA: the synthesis engine's density).
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
for nf in 0 18 36 54; do for occ in 1 2; do ./tools/probes/occ_${occ}_${nf} 512; done; done 2>&1 | tee $OUT/two_waves_time.txt
cd /tmp && export TMPDIR=/tmp
for occ in 1 2; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_occ$occ -o occ$occ -- $GRAFT_REPO_ROOT/tools/probes/occ_${occ}_36 512 > $OUT/stats_occ$occ.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc1_occ$occ -o occ$occ -- $GRAFT_REPO_ROOT/tools/probes/occ_${occ}_36 512 > $OUT/pmc1_occ$occ.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $OUT/pmc2_occ$occ -o occ$occ -- $GRAFT_REPO_ROOT/tools/probes/occ_${occ}_36 512 > $OUT/pmc2_occ$occ.log 2>&1
done
find $OUT -name "*.csv" | head -20
