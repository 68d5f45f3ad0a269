#!/bin/bash
# After `gpurun -- 'bash tools/round3_artifacts.sh'` (and, for the bench line's PMC traffic, one more
# `gpurun -- 'python bench.py > gpurun_out/r3_final/bench_n1_65536.json'` once profiles/pmc_latest.json is in place):
# turn gpurun_out/ into the files committed under profiles/.
set -e
python tools/pmc_summary.py r3_final
for f in bench_131072_config3_shard bench_all_hints_65536 bench_flow_65536 bench_total_1048576_1gpu witness_level10_1 witness_level10_256 \
         witness_level10_1024 witness_level10_4096 witness_level10_16384 witness_level10_16384_by_variable witness_level1_1024 witness_rec16_x5_1024; do
    cp gpurun_out/r3_art/$f.json profiles/r3_$f.json
done
cp gpurun_out/r3_art/bench_matrix.txt profiles/r3_bench_matrix.txt
cp gpurun_out/r3_art/perm_census.txt profiles/r3_perm_census.txt
cp gpurun_out/r3_art/soak.txt profiles/r3_soak.txt
cp gpurun_out/r3_art/witness_kt/kt_kernel_stats.csv profiles/r3_witness_kernel_stats_4096.csv
cp gpurun_out/r3_final/bench_n1_65536.json profiles/r3_final_bench_n1_65536.json
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
p = json.load(open("profiles/pmc_latest.json"))
d = json.load(open("profiles/r3_final_bench_n1_65536.json"))
print("kernel sources", bench.kernel_sources_sha(), "profiled", p["kernel_sources_sha"])
print("bench", round(d["value"]), "proofs/s", round(d["ms_per_step"], 2), "ms, traffic", d["roofline"]["traffic"])
PY
