#!/bin/bash
# Round 6: the whole GPU suite on the current tree, then the same-lease A/B of persistent workgroups in field_x3t (H3D_FIELD_X3T_WG_PER_CU=0:
# one 64-sample group per workgroup, as rounds 2-5 launched it).
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
for rep in 1 2; do for per in 0 8; do
  H3D_FIELD_X3T_WG_PER_CU=$per timeout 300 python bench.py --config MAP3DBN512L --no-cpu --no-extra --no-check --steps 5 --warmup 2 > $OUT/L_${per}_$rep.json 2> $OUT/L_${per}_$rep.err
  H3D_FIELD_X3T_WG_PER_CU=$per timeout 300 python bench.py --config MAP3DBN --res 256x256 --render 64x64 --samples 32 --batch 8 --no-cpu --no-extra --no-check --steps 20 --warmup 5 > $OUT/2_${per}_$rep.json 2> $OUT/2_${per}_$rep.err
  python - <<PY
import json
for k in ("L","2"):
    try:
        d=json.loads(open("$OUT/%s_${per}_$rep.json" % k).read().strip().split("\n")[-1])
        print(k, "wg_per_cu=$per", "$rep", d["value"], d["ms_per_step"], d.get("stage_ms"))
    except Exception as e:
        print(k, "$per $rep failed", e, open("$OUT/%s_${per}_$rep.err" % k).read()[-600:])
PY
done; done 2>&1 | tee $OUT/persist_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
