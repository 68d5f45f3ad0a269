#!/bin/bash
# Everything profiles/r5_* is made from, on the GPU box, in two calls (each within gpurun's limit), on ONE set of kernel sources:
#   gpurun --timeout 1150 -- 'bash tools/round5_artifacts.sh a'     tests, profile (kernel trace + 4 PMC passes), bench matrix, sweeps, side legs
#   gpurun --timeout 1150 -- 'bash tools/round5_artifacts.sh b'     census, timelines, the soaks — the LARGE soak LAST
# then, here:  python tools/pmc_summary.py r5_final; python tools/round5_collect.py
# tests/soak.py prints the hash of the kernel sources it ran on; tests/test_bench_contract.py holds it against pmc_latest.json's.
PART=${1:-a}
OUT=gpurun_out/r5_art
mkdir -p $OUT
if [ "$PART" = a ]; then
python -m pytest tests -m gpu -x -q --durations=6 > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -4 $OUT/tests.log
bash tools/profile.sh r5_final 2>&1 | grep -E "done|Error|error"
bash tools/bench_matrix.sh r5_art/matrix > $OUT/bench_matrix.txt 2>&1; cat $OUT/bench_matrix.txt
bash tools/sweep.sh r5_art/sweep "1 16 128 256 512 1024 2048 4096 8192 16384 32768 65536" default >> $OUT/bench_matrix.txt 2>&1; tail -12 $OUT/bench_matrix.txt
timeout -k 10 300 python bench.py --workload copies --proofs 65536 --emit-flow --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_flow_65536.json 2> $OUT/flow.err; echo flow done
timeout -k 10 300 python bench.py --workload copies --proofs 65536 --emit-paths --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_all_hints_65536.json 2> $OUT/hints.err; echo hints done
timeout -k 10 300 python bench.py --proofs 131072 --steps 3 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_131072_config3_shard.json 2> $OUT/shard.err; echo shard done
timeout -k 10 500 python bench.py --total-proofs 1048576 --steps 1 --warmup 1 --cpu-sample 0 --perm-log2 0 > $OUT/bench_total_1048576_1gpu.json 2> $OUT/total.err; echo total done
# the C-ABI's multi-GPU paths under the same clock, and the level-ordered chain cut by bytes (#2)
timeout -k 10 300 python bench.py --exchange c --steps 5 --warmup 1 --cpu-sample 0 --perm-log2 0 --no-single-proof > $OUT/bench_exchange_c_65536.json 2> $OUT/xc.err; echo exchange-c done
timeout -k 10 300 python bench.py --devices 0 --steps 5 --warmup 1 > $OUT/bench_devices_0_65536.json 2> $OUT/dev0.err; echo devices-0 done
timeout -k 10 300 python bench.py --devices 0,0 --workload chain --order level --total-proofs 53248 --steps 2 --warmup 1 > $OUT/bench_devices_00_chain_level_53248.json 2> $OUT/dev00.err; echo devices-0,0 done
timeout -k 10 300 python tools/host_path_bench.py 10240 > $OUT/host_path_10240.json 2> $OUT/host.err; echo host10240 done
timeout -k 10 300 python tools/host_path_bench.py 32768 > $OUT/host_path_32768.json 2>> $OUT/host.err; echo host32768 done
for N in 1 1024 16384; do timeout -k 10 250 python tools/bench_witness.py --fixture level10-1.bin --proofs $N > $OUT/witness_level10_$N.json 2> $OUT/witness.err || tail -3 $OUT/witness.err; done; echo witness done
python3 - <<'PY'
import json
for f in ("bench_flow_65536","bench_all_hints_65536","bench_131072_config3_shard","bench_total_1048576_1gpu","bench_exchange_c_65536","bench_devices_0_65536","bench_devices_00_chain_level_53248","witness_level10_1","witness_level10_1024","witness_level10_16384"):
    try:
        d=json.loads([l for l in open("gpurun_out/r5_art/%s.json"%f) if l.startswith("{")][-1]); print(f, round(d["value"]), round(d["ms_per_step"],2))
    except Exception as e: print(f,"FAILED",e)
PY
else
bash tools/timeline_small.sh r5_art/tl "1 128 1024" > $OUT/timelines.txt 2>&1; echo timelines done
bash tools/valu_clock.sh r5_lab > $OUT/lab.log 2>&1; echo lab done
PERM_FORM_LIST=0,1,2 PERM_WG_LIST=6,8,12,16,24,32 timeout -k 10 200 python tools/perm_bench.py 24 > $OUT/perm_bench.txt 2>&1; echo perm done
timeout -k 10 300 python tests/perm_census.py chain > $OUT/perm_census.txt 2>&1 && echo census done || { echo "census FAILED"; tail -3 $OUT/perm_census.txt; }
timeout -k 10 300 python tests/soak.py 1500 41 - single 300 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
timeout -k 10 300 python tests/soak.py 1500 42 pow0 >> $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
# every kernel form forced over a smaller corpus (one-configuration calls reach the row forms)
timeout -k 10 300 python tests/soak.py 300 51 - single 150 cap_mid=on cap_top=on tree_pace=paced > $OUT/soak_forms.txt 2>&1; tail -1 $OUT/soak_forms.txt
timeout -k 10 300 python tests/soak.py 300 52 pow0 single 150 cap_mid=off oods_form=lane transcript_form=lane >> $OUT/soak_forms.txt 2>&1; tail -1 $OUT/soak_forms.txt
# LAST: the large soak on these sources
timeout -k 10 500 python tests/soak.py 3000 61 - single 1200 > $OUT/soak_large.txt 2>&1; tail -3 $OUT/soak_large.txt
timeout -k 10 500 python tests/soak.py 3000 62 pow0 single 1200 >> $OUT/soak_large.txt 2>&1; tail -3 $OUT/soak_large.txt
fi
