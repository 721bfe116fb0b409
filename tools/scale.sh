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
    if [ "$N" -eq 1 ]; then
      python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --mode $MODE $EXTRA > "$OUT/${MODE}_n$N.json" 2> "$OUT/${MODE}_n$N.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
        bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --mode $MODE $EXTRA > "$OUT/${MODE}_n$N.json" 2> "$OUT/${MODE}_n$N.err"
    fi
    tail -c 300 "$OUT/${MODE}_n$N.json"; echo
  done
done
