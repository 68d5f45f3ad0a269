#!/bin/bash
# A/B of one knob over the side workloads:  gpurun -- 'bash tools/bench_ab.sh <tag> <knob=value> [<knob=value> ...]'
# Every workload is run without knobs and then once per given knob setting.
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --cpu-sample 0 --perm-log2 0 > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); s=d["roofline"]["stage_ms"]
    print("$name", round(d["value"]), "proofs/s", round(d["ms_per_step"],3), "ms", {k:round(v,3) for k,v in s.items() if v})
except Exception as e: print("$name FAILED", e)
PY
}
for K in "" "$@"; do
    KN=${K:+--knob $K}; L=${K:-default}
    run standard_65536_$L --steps 4 --warmup 1 $KN
    run copies_1024_$L --workload copies --proofs 1024 --steps 30 --warmup 5 $KN
    run copies_4096_$L --workload copies --proofs 4096 --steps 15 --warmup 3 $KN
    run copies_16384_$L --workload copies --proofs 16384 --steps 6 --warmup 2 $KN
    run chain_65536_$L --workload chain --proofs 65536 --steps 3 --warmup 1 $KN
done
