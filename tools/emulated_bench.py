"""Emulated-Poseidon2 gate-value microbenchmark: rsv_poseidon2_emulated_dev on 2^k permutations resident
in HBM.  The kernel is HBM-write bound: 65 B in + 6 656 B out per permutation (52 whole 128-byte lines).
python tools/emulated_bench.py [log2_perms] [json_out]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402

rsv = rsvload.load_package()
import torch  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
d_l = torch.randint(0, 0x7FFFFFFF, (n, 8), dtype=torch.int32, device=dev, generator=gen)
d_r = torch.randint(0, 0x7FFFFFFF, (n, 8), dtype=torch.int32, device=dev, generator=gen)
d_s = torch.randint(0, 3, (n,), dtype=torch.uint8, device=dev, generator=gen)
d_rows = torch.empty((n, rsv.EMU_STRIDE, 4), dtype=torch.int32, device=dev)
ctx = rsv.Context(0)
ctx.poseidon2_emulated(d_l, d_r, d_s, d_rows)
ctx.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.poseidon2_emulated(d_l, d_r, d_s, d_rows)
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) / 5)
bytes_per = 64 + 1 + rsv.EMU_STRIDE * 16
t0 = time.perf_counter()
for _ in range(5):
    d_rows.fill_(1)
torch.cuda.synchronize()
fill = (time.perf_counter() - t0) / 5
res = {"fill_GBps": d_rows.numel() * 4 / fill / 1e9, "perms": n, "ms": best * 1e3, "perms_per_s": n / best, "algorithmic_bytes_per_perm": bytes_per,
       "GBps": n * bytes_per / best / 1e9, "frac_of_8TBps": n * bytes_per / best / 8e12}
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
