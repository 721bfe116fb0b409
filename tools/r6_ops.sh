#!/bin/bash
# HBM-bound plugin ops: shipped library vs a build with non-temporal output stores, same lease; then FETCH / WRITE counters of the shipped one.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
for rep in 1 2; do for lib in libh3d.so libh3d_nt.so; do echo "== $lib $rep"; H3D_LIB=$C/$lib timeout 300 python tools/op_rooflines.py 2>/dev/null | grep -v conv; done; done | tee $OUT/ops_ab.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/op_rooflines.py > /dev/null 2> $OUT/pmc_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py $(find $OUT/pmc_$c -name '*.db' | head -1) "upfirdn2d|bilinear|bias_act" > $OUT/pmc_$c.txt
done
find $OUT -name '*.db' -delete
cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
