#!/bin/bash
# Batch-size sweep of the copies workload under knob settings:  gpurun -- 'bash tools/sweep.sh <tag> "<sizes>" <knob=value|default> ...'
TAG=$1; SIZES=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
for K in "$@"; do
  for N in $SIZES; do
    KN=""; [ "$K" != "default" ] && KN="--knob $K"
    ST=$(( 40000 / N + 4 )); [ $ST -gt 40 ] && ST=40
    timeout -k 10 200 python bench.py --workload copies --proofs $N --steps $ST --warmup 4 --cpu-sample 0 --perm-log2 0 --no-stage-times --no-single-proof $KN > $OUT/c_${N}_$K.json 2> $OUT/c_${N}_$K.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/c_${N}_$K.json")); print("$K", $N, round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms")
except Exception as e: print("$K $N FAILED", e)
PY
  done
done
