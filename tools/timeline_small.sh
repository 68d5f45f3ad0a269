#!/bin/bash
# Kernel timeline of one small-batch step:  gpurun -- 'bash tools/timeline_small.sh <tag> "<sizes>"'
TAG=$1; SIZES=${2:-"1 1024"}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for N in $SIZES; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$N -o kt -- python3 $ROOT/bench.py --workload copies --proofs $N --steps 6 --warmup 3 --cpu-sample 0 --perm-log2 0 --no-single-proof --no-stage-times > $OUT/kt_$N.json 2> $OUT/kt_$N.err
  F=$(find $OUT/kt_$N -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/timeline.py $F > $OUT/timeline_$N.txt 2>&1
  echo "== $N"; cat $OUT/timeline_$N.txt
  find $OUT/kt_$N -name "*.csv" -size +4M -delete
done
