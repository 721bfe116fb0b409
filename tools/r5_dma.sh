#!/bin/bash
# Same-lease A/B of the place of the ring's DMA piece inside a GEMM section (H3D_DMA_SLOT: 0 after the MFMAs (default), 1 top of
# the section, 2 after the fragment reads, 3 after the producer hook).  usage: bash tools/r5_dma.sh
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r5d
mkdir -p $OUT
for rep in 1 2; do for lib in libh3d.so libh3d_dma1.so libh3d_dma2.so libh3d_dma3.so; do
  name=$(basename $lib .so)_$rep
  chk="--no-check"; [ $rep = 1 ] && chk="--check-items 2"
  H3D_LIB=$PWD/3dhumangan_amd/csrc/$lib timeout 300 python bench.py --no-cpu --no-extra $chk --steps 20 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().split("\n")[-1])
    c=d.get("checked") or {}
    print("$name", d["value"], d["ms_per_step"], d["stage_ms"]["render_fused"], d["stage_ms"]["synthesis"], c.get("max_rel_err"), c.get("max_rel_err_render"), c.get("x2_fell_back"))
except Exception as e:
    print("$name failed", e)
PY
done; done
