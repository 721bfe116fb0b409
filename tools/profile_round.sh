#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats of the same command, and separate PMC passes for HBM traffic.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r1
set -u
R=${1:-r1}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
REPO=$PWD
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $REPO/bench.py --no-cpu --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p -- python $REPO/bench.py --no-cpu --no-extra --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$c.err
done
cd $REPO
DB=$(find $OUT/stats -name '*.db' | head -1)
python tools/rocprof_summary.py $DB $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  DB=$(find $OUT/pmc_$c -name '*.db' | head -1)
  python tools/pmc_dump.py $DB 'x3_kernel|geo_features|ray_integrate' > $OUT/pmc_$c.txt
done
python tools/traffic_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt MAP3DBN512_512x512_b16_s64 $OUT/hbm_traffic.json
find $OUT -name '*.db' -size +12M -delete; ls -la $OUT/*/*/* 2>/dev/null | head
tail -c 600 $OUT/bench.json
