#!/bin/bash
# experiment: non-temporal output stores in h3d_conv_x3 -- cfg 4 iteration (fp32, AMP, AMP with the own 1x1 f16 kernel for the dense layers), same lease
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
C=$PWD/3dhumangan_amd/csrc
for lib in libh3d.so libh3d_convnt.so; do
  for mode in "fp32 --amp none" "amp --amp fp16"; do set -- $mode; name=$1; shift
    H3D_LIB=$C/$lib timeout 300 python bench.py --mode trainstep --batch 4 --steps 5 --warmup 6 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$name', round(d['ms_per_step'],2), d['stage_ms'])"
  done
  H3D_AMP_LINEAR=x3 H3D_LIB=$C/$lib timeout 300 python bench.py --mode trainstep --batch 4 --steps 5 --warmup 6 --amp fp16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'amp_linear_x3', round(d['ms_per_step'],2), d['stage_ms'])"
done | tee $OUT/convnt.txt
