#!/usr/bin/env python3
"""gpurun_out/r5_art + gpurun_out/r5_lab (tools/round5_artifacts.sh a / b) -> profiles/r5_*.  Usage: python tools/round5_collect.py"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r5_art"), os.path.join(ROOT, "profiles")
for name in ("bench_131072_config3_shard.json", "bench_all_hints_65536.json", "bench_flow_65536.json", "bench_total_1048576_1gpu.json",
             "bench_exchange_c_65536.json", "bench_devices_0_65536.json", "bench_devices_00_chain_level_53248.json",
             "bench_matrix.txt", "host_path_10240.json", "host_path_32768.json", "perm_census.txt", "perm_bench.txt", "soak.txt", "soak_forms.txt",
             "soak_large.txt", "witness_level10_1.json", "witness_level10_1024.json", "witness_level10_16384.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "r5_" + name))
    else:
        print("missing", name)
for n in ("1", "128", "1024"):
    p = os.path.join(src, "tl", f"timeline_{n}.txt")
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(dst, f"r5_timeline_{n}.txt"), "w") as g:
            g.write(f"# rocprofv3 --kernel-trace of bench.py --workload copies --proofs {n} --no-stage-times (last step; tools/timeline_small.sh, tools/timeline.py), us\n")
            g.write(f.read())
lab = os.path.join(ROOT, "gpurun_out", "r5_lab")
for name in ("valu_lab_1waves.txt", "valu_lab_4waves.txt", "valu_lab_8waves.txt", "valu_lab_grbm_clock.txt"):
    p = os.path.join(lab, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "r5_" + name))
with open(os.path.join(src, "tests.log")) as f, open(os.path.join(dst, "r5_gpu_tests.txt"), "w") as g:
    g.write("# python -m pytest tests -m gpu -x -q --durations=6 on the GPU box (tools/round5_artifacts.sh a)\n" + f.read())
print("copied")
