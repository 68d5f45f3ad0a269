#!/bin/bash
# A/B of several BUILDS of the library on one box, alternating (copies workload):  gpurun -- 'bash tools/ab_libs.sh <tag> "<sizes>" default <other.so> ...'
TAG=$1; SIZES=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for R in 1 2 3; do
for N in $SIZES; do
  for L in "$@"; do
    LA=""; [ "$L" != default ] && LA="--lib $L"
    ST=$(( 40000 / N + 4 )); [ $ST -gt 40 ] && ST=40
    name=$(basename $L .so)_${N}_$R
    timeout -k 10 200 python bench.py --workload copies --proofs $N --steps $ST --warmup 4 --cpu-sample 0 --perm-log2 0 --no-stage-times --no-single-proof $LA > $OUT/$name.json 2> $OUT/$name.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); print("$name", round(d["ms_per_step"],3), "ms")
except Exception as e: print("$name FAILED", e)
PY
  done
done
done
