#!/bin/bash
# Regenerates the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile.sh r3_final'
# Outputs under gpurun_out/<tag>/ ; tools/pmc_summary.py turns them into the files committed under profiles/.
# rocprofv3: the program itself follows `--` (no env/bash hop), --pmc passes are separate runs without trace domains.
set -e -o pipefail
TAG=${1:-r3_final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# --witness-proofs 0 --no-single-proof: the side legs re-run the pipeline's kernels on a small batch, and the summaries take each kernel's LAST dispatch
BENCH_ARGS="--steps 2 --warmup 1 --cpu-sample 0 --perm-log2 20 --witness-proofs 0 --no-single-proof"
timeout -k 10 400 python3 bench.py > "$OUT/bench_n1_65536.json" 2> "$OUT/bench.err"
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
echo "kernel trace done"
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_sources_sha())" > "$OUT/kernel_sources_sha.txt"
# four separate --pmc passes (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ: 8 slots; GRBM on its own)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    NAME=$(echo $C | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-sample 0 --perm-log2 20 --witness-proofs 0 --no-single-proof > "$OUT/pmc_$NAME.json" 2> "$OUT/pmc_$NAME.err"
    echo "pmc $NAME done"
done
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete   # keep the merge under gpurun's 64 MiB limit
ls -R "$OUT" | head -40
