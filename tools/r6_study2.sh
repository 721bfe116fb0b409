#!/bin/bash
cd "$(dirname "$0")/.."
C=$PWD/3dhumangan_amd/csrc
mkdir -p gpurun_out/r6s
H3D_LIB=$C/libh3d_tracefine.so timeout 300 python tools/synth_x3t_trace.py MAP3DBN > gpurun_out/r6s/trace_synth_384_x2t_fine.txt 2>&1
python tools/x2_oracle_study.py > gpurun_out/r6s/x2_oracle_study.jsonl 2> gpurun_out/r6s/x2_oracle_study.err; tail -c 1200 gpurun_out/r6s/x2_oracle_study.jsonl
