#!/bin/bash
# A/B of two builds of the library on ONE box, alternating:  gpurun -- 'bash tools/ab_lib.sh <tag> <other.so> [rounds] [bench args]'
TAG=$1; OTHER=$2; ROUNDS=${3:-3}; shift 3
OUT=gpurun_out/$TAG
mkdir -p $OUT
for R in $(seq 1 $ROUNDS); do
  for W in this other; do
    L=""; [ $W = other ] && L="--lib $OTHER"
    timeout -k 10 300 python bench.py --cpu-sample 0 --perm-log2 0 --no-single-proof $L "$@" > $OUT/${W}_$R.json 2> $OUT/${W}_$R.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/${W}_$R.json")); s=d["roofline"]["stage_ms"] or {}
    print("$W", $R, round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms", {k:round(v,2) for k,v in s.items() if v})
except Exception as e: print("$W $R FAILED", e)
PY
  done
done
