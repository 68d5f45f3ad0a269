#!/usr/bin/env python3
"""gpurun_out/r4_art (tools/round4_artifacts.sh) -> profiles/r4_*.  Usage: python tools/round4_collect.py"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r4_art"), os.path.join(ROOT, "profiles")
for name in ("bench_131072_config3_shard.json", "bench_all_hints_65536.json", "bench_flow_65536.json", "bench_total_1048576_1gpu.json",
             "bench_matrix.txt", "host_path_10240.json", "host_path_32768.json", "perm_census.txt", "soak.txt",
             "witness_level10_1.json", "witness_level10_1024.json", "witness_level10_16384.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "r4_" + name))
for n in ("1", "128", "1024"):
    with open(os.path.join(src, "tl", f"timeline_{n}.txt")) as f, open(os.path.join(dst, f"r4_timeline_{n}.txt"), "w") as g:
        g.write(f"# rocprofv3 --kernel-trace of bench.py --workload copies --proofs {n} --no-stage-times (last step; tools/timeline_small.sh, tools/timeline.py), us\n")
        g.write(f.read())
with open(os.path.join(src, "tests.log")) as f, open(os.path.join(dst, "r4_gpu_tests.txt"), "w") as g:
    g.write("# python -m pytest tests -m gpu -x -q --durations=6 on the GPU box (tools/round4_artifacts.sh)\n" + f.read())
print("copied")
