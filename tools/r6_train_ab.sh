#!/bin/bash
# config-4 iteration A/B in one lease.  usage (GPU box, repo root): bash tools/r6_train_ab.sh <tag> "<ENV=a>" "<ENV=b>" [amp]
TAG=$1; A=$2; B=$3; AMP=${4:-none}
OUT=gpurun_out
W=2; [ "$AMP" = fp16 ] && W=6
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $v python bench.py --mode trainstep --batch 4 --steps 8 --warmup $W --amp $AMP 2>/dev/null | tail -1 > $OUT/${TAG}_tmp.json
    python - "$v" $rep $OUT/${TAG}_tmp.json <<'PY' | tee -a $OUT/${TAG}.txt
import json, sys
d = json.loads(open(sys.argv[3]).read())
print(sys.argv[1], "rep", sys.argv[2], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms"].items()}, "d_loss", d["loss"]["d_loss"], "g_loss", d["loss"]["g_loss"])
PY
  done
done
