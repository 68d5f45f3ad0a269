#!/usr/bin/env python3
"""One line per kernel from `make -C recursive-stwo_amd/csrc usage` (hipcc -Rpass-analysis=kernel-resource-usage):
name, VGPRs, SGPRs, scratch bytes/lane, occupancy (waves/SIMD), LDS bytes/block."""
import re
import subprocess
import sys

out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                      "-c", "-o", "/dev/null", "rsv_hip.hip"] + sys.argv[1:], cwd="recursive-stwo_amd/csrc", capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':60s} {'VGPR':>5s} {'SGPR':>5s} {'scr':>5s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('vgpr', -1):5d} {r.get('sgpr', -1):5d} {r.get('scratch', -1):5d} {r.get('occ', -1):4d} {r.get('lds', -1):6d}")
