#!/bin/bash
# Scaling runs on ONE node: bench.py at 1 / 2 / 4 / 8 GPUs for the generator benchmark (replicas, no data-path collective) and
# for the config-4 training iteration (SyncBN moment all-reduces, R1 all-gather, bucketed gradient all-reduce over RCCL).
# One JSON line per N under <outdir>; efficiency is NOT computed here (the driver computes it from the per-N values).
# usage: tools/scale.sh [outdir] [steps] [warmup]
set -u
OUT=${1:-gpurun_out/scale}; STEPS=${2:-20}; WARM=${3:-5}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0            # dmabuf IPC only on this driver (RCCL / tensor sharing across processes)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && { echo "skipping N=$N: $NG GPUs visible"; continue; }
  PORT=$((29500 + N))
  for MODE in generator trainstep; do
    EXTRA="--no-cpu --no-extra"; [ "$MODE" = trainstep ] && EXTRA="--batch 4"
    # bench.py re-executes itself under torch.distributed.run for N > 1 (one rank per GPU); its last stdout line is the compact
    # contract line (< 4 KB), the full record goes to bench_detail.json
    python bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --mode $MODE $EXTRA > "$OUT/${MODE}_n$N.json" 2> "$OUT/${MODE}_n$N.err"
    [ -f bench_detail.json ] && [ "$MODE" = generator ] && cp bench_detail.json "$OUT/${MODE}_n${N}_detail.json"
    tail -n 1 "$OUT/${MODE}_n$N.json" | cut -c 1-400; echo
  done
done
