#!/bin/bash
# GPU box: parity of the x3t engines + their throughput on the wide BASELINE workloads.  usage: tools/x3t_bench.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_generator.py -x -q -k "x3t" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "256 f16x3" "256 f16x3t" "384 f16x3t" "420 f16x3t"; do set -- $cfg; echo -n "fused $1 $2: "; timeout 300 python tools/microbench.py --what fused --B 8 --R 9216 --S 64 --F $1 --engine $2 2>&1 | tail -1; done | tee $O/micro.log
python bench.py --config MAP3DBN512L --no-cpu --no-extra --no-check --steps 5 > $O/b_3L.json 2> $O/b_3L.err
python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --no-check --steps 10 > $O/b_c2.json 2> $O/b_c2.err
H3D_FIELD_PRECISION=f16x3t H3D_SYNTH_PRECISION=bf16x3t python bench.py --no-cpu --no-extra --no-check --steps 5 > $O/b_c3_x3t.json 2> $O/b_c3_x3t.err
python - <<EOF2
import json
for f in ("b_3L","b_c2","b_c3_x3t"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms", d["stage_ms"])
    except Exception as e: print(f, "ERR", e, open("$O/%s.err"%f).read()[-800:])
EOF2
