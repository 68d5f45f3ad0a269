#!/bin/bash
# Everything profiles/r4_* is made from, on the GPU box:  gpurun --timeout 1150 -- 'bash tools/round4_artifacts.sh'
# (then, here: python tools/pmc_summary.py r4_final; python tools/round4_collect.py)
OUT=gpurun_out/r4_art
mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=6 > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -4 $OUT/tests.log
bash tools/profile.sh r4_final 2>&1 | grep -E "done|Error|error"
bash tools/bench_matrix.sh r4_art/matrix > $OUT/bench_matrix.txt 2>&1; cat $OUT/bench_matrix.txt
bash tools/sweep.sh r4_art/sweep "1 16 128 256 512 1024 2048 4096 8192 16384 32768 65536" default >> $OUT/bench_matrix.txt 2>&1; tail -11 $OUT/bench_matrix.txt
bash tools/sweep.sh r4_art/sweep "1024 2048" pair_order=off >> $OUT/bench_matrix.txt 2>&1; tail -2 $OUT/bench_matrix.txt
bash tools/sweep.sh r4_art/sweep "1 128 256 512" tree_pace=unpaced >> $OUT/bench_matrix.txt 2>&1; tail -4 $OUT/bench_matrix.txt
timeout -k 10 300 python bench.py --workload copies --proofs 65536 --emit-flow --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_flow_65536.json 2> $OUT/flow.err; echo flow done
timeout -k 10 300 python bench.py --workload copies --proofs 65536 --emit-paths --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_all_hints_65536.json 2> $OUT/hints.err; echo hints done
timeout -k 10 300 python bench.py --proofs 131072 --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_131072_config3_shard.json 2> $OUT/shard.err; echo shard done
timeout -k 10 500 python bench.py --total-proofs 1048576 --steps 1 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_total_1048576_1gpu.json 2> $OUT/total.err; echo total done
timeout -k 10 300 python tools/host_path_bench.py 10240 > $OUT/host_path_10240.json 2> $OUT/host.err; echo host10240 done
timeout -k 10 300 python tools/host_path_bench.py 32768 > $OUT/host_path_32768.json 2>> $OUT/host.err; echo host32768 done
bash tools/timeline_small.sh r4_art/tl "1 128 1024" > $OUT/timelines.txt 2>&1; echo timelines done
for N in 1 1024 16384; do timeout -k 10 250 python tools/bench_witness.py --fixture level10-1.bin --proofs $N > $OUT/witness_level10_$N.json 2> $OUT/witness.err || tail -3 $OUT/witness.err; done; echo witness done
timeout -k 10 300 python tests/perm_census.py > $OUT/perm_census.txt 2>&1 && echo census done || { echo "census FAILED"; tail -3 $OUT/perm_census.txt; }
timeout -k 10 300 python tests/soak.py 1500 41 - single 300 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
timeout -k 10 300 python tests/soak.py 1500 42 pow0 >> $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
python3 - <<'PY'
import json
for f in ("bench_flow_65536","bench_all_hints_65536","bench_131072_config3_shard","bench_total_1048576_1gpu","witness_level10_1","witness_level10_1024","witness_level10_16384"):
    try:
        d=json.load(open("gpurun_out/r4_art/%s.json"%f)); print(f, round(d["value"]), round(d["ms_per_step"],2))
    except Exception as e: print(f,"FAILED",e)
PY
