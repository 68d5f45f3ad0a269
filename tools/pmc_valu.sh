#!/bin/bash
# VALU-busy PMC pass: per-kernel SQ busy / wait / active counters of one bench step.
#   gpurun -- 'bash tools/pmc_valu.sh <tag> [RSV_EXP value]'
# rocprofv3: program directly after `--`, --pmc alone (no trace domains).
set -e -o pipefail
TAG=${1:-r2_valu}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$2" ] && export RSV_EXP=$2
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
grep -oE "SQ_[A-Z_0-9]+" "$OUT/counters_list.txt" | sort -u > "$OUT/sq_counters.txt" || true
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "GRBM_GUI_ACTIVE"; do
    NAME=$(echo $C | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-sample 0 --perm-log2 20 > "$OUT/pmc_$NAME.json" 2> "$OUT/pmc_$NAME.err" || echo "pass $NAME failed"
    echo "pmc $NAME done"
done
ls -R "$OUT" | head -30
