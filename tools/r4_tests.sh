#!/bin/bash
# usage: tools/r4_tests.sh <tag> <pytest args...>   -> gpurun_out/<tag>.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
timeout 1500 python -m pytest "$@" -q -m gpu -s -p no:cacheprovider > gpurun_out/${tag}.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}.log
grep -E "passed|failed|error|rc=|^FAILED|fell back|engine " gpurun_out/${tag}.log | tail -40
