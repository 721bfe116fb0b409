#!/bin/bash
# usage: bash tools/pmc_pass.sh <tag> "<counters>" -- <python args...>   (one rocprofv3 --pmc pass; prints per-kernel sums)
TAG=$1; CTRS=$2; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS -d $OUT -o p -- python "$@" > $OUT/run.log 2> $OUT/run.err
cd $REPO
DB=$(find $OUT -name '*.db' | head -1)
python tools/pmc_dump.py $DB 'x3_kernel|x3t_kernel|geo_features' | tee $OUT/pmc.txt
rm -f $DB
