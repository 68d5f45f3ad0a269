#!/bin/bash
# Build tools/perm_lab and print the instruction mix of the optimized permutation kernel.
set -e
cd "$(dirname "$0")"
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../recursive-stwo_amd/csrc -o perm_lab perm_lab.hip 2>&1 | grep -E "error" -A5 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../recursive-stwo_amd/csrc --cuda-device-only -S -o /tmp/asm/lab.s perm_lab.hip 2>/dev/null
python3 - <<'PY'
import re,collections
s=open('/tmp/asm/lab.s').read()
funcs=re.split(r'\n(?=\S+:\s*; @)',s)
for f in funcs:
    m=re.match(r'(\S+):',f)
    if not m or 'k_permILi1' not in m.group(1): continue
    ins=[l.strip().split()[0] for l in f.split('\n') if l.startswith('\t') and not l.strip().startswith(('.',';','//'))]
    c=collections.Counter(ins)
    print(m.group(1),len(ins)); print(c.most_common(24))
PY
grep -E "vgpr_count|scratch_en|private_segment_fixed_size" /tmp/asm/lab.s | head
