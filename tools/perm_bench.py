"""Poseidon2 permutation microbenchmark (BASELINE.json's second metric): rsv_poseidon2_permute_dev on 2^k states resident
in HBM, HIP-event timed on the context's stream.  python tools/perm_bench.py [log2_states] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402

rsv = rsvload.load_package()
import torch  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = 1 << k
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
d_in = torch.randint(0, 0x7FFFFFFF, (m, 16), dtype=torch.int32, device=dev, generator=gen)
d_out = torch.empty_like(d_in)
ctx = rsv.Context(0)
for form, wg in [(f, w) for f in os.environ.get("PERM_FORM_LIST", "0").split(",") for w in os.environ.get("PERM_WG_LIST", "24").split(",")]:
    ctx.set_option("perm_form", int(form))
    ctx.set_option("perm_wg_per_cu", int(wg))
    ctx.poseidon2_permute(d_in, d_out)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.poseidon2_permute(d_in, d_out)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    print(f"form={form} wg_per_cu={wg}: {m / best / 1e9:.3f} G perms/s, {m * 128 / best / 1e9:.0f} GB/s")
