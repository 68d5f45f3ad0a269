#!/bin/bash
# Side workloads of DESIGN §5 (parity-test configurations, not the bench line):  gpurun -- 'bash tools/bench_matrix.sh <tag>'
TAG=${1:-r2_matrix}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --cpu-sample 0 --perm-log2 0 > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d["roofline"]["stage_ms"]
    print("$name", round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms", {k:round(v,3) for k,v in s.items()})
except Exception as e: print("$name FAILED", e)
PY
}
run standard_65536 --steps 3 --warmup 1
run copies_1024 --workload copies --proofs 1024 --steps 20 --warmup 3
run copies_4096 --workload copies --proofs 4096 --steps 10 --warmup 2
run copies_16384 --workload copies --proofs 16384 --steps 5 --warmup 2
run chain_16384 --workload chain --proofs 16384 --steps 3 --warmup 1
run chain_65536 --workload chain --proofs 65536 --steps 3 --warmup 1
