#!/bin/bash
# Round 6 closing lease on the final tree: full GPU suite, smoke(), the oracle study of the shipped default, the profile round.
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r6z; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
timeout 1500 python tools/x2_oracle_study.py > $OUT/x2_oracle_study.jsonl 2> $OUT/x2_oracle_study.err; tail -c 900 $OUT/x2_oracle_study.jsonl
bash tools/r6_profile.sh r6z
