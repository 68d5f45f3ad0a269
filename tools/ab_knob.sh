#!/bin/bash
# A/B of knob settings over chosen workloads on ONE box, alternating:
#   gpurun -- 'bash tools/ab_knob.sh <tag> "<workload:proofs:steps> ..." <knob=value|default> [<knob=value> ...]'
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
for R in 1 2; do
for W in $WL; do
  IFS=: read -r wl n steps <<< "$W"
  for K in "$@"; do
    KN=""; [ "$K" != default ] && KN="--knob $K"
    name=${wl}_${n}_${K}_$R
    timeout -k 10 300 python bench.py --workload $wl --proofs $n --steps $steps --warmup 1 --cpu-sample 0 --perm-log2 0 --no-single-proof $KN > $OUT/$name.json 2> $OUT/$name.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d["roofline"]["stage_ms"] or {}
    print("$name", round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms", {k:round(v,2) for k,v in s.items() if v})
except Exception as e: print("$name FAILED", e)
PY
  done
done
done
