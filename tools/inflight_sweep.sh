#!/bin/bash
# One batch of N proofs per step against K batches of N / K in flight on K contexts of the same GPU (bench.py --inflight K):
#   gpurun -- 'bash tools/inflight_sweep.sh <tag> "<total sizes>"'
TAG=$1; SIZES=${2:-"2048 4096 8192 16384 32768 65536"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for N in $SIZES; do
  for K in 1 2 4; do
    P=$(( N / K )); ST=$(( 60000 / N + 4 )); [ $ST -gt 30 ] && ST=30
    timeout -k 10 200 python bench.py --workload copies --proofs $P --inflight $K --steps $ST --warmup 3 --cpu-sample 0 --perm-log2 0 --no-stage-times --no-single-proof > $OUT/i_${N}_$K.json 2> $OUT/i_${N}_$K.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/i_${N}_$K.json")); print("total $N in flight $K:", round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms per step of $N")
except Exception as e: print("$N $K FAILED", e)
PY
  done
done
