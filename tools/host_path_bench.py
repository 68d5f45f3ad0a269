"""End-to-end rate when the proofs start in HOST memory (PCIe inclusive): rsv_verify_batch_host (pipelined
gather / upload / verify) against the plain rsv_verify_batch (one blob, synchronous upload).  Not the bench line:
bench.py's value is measured with inputs resident in HBM.  Usage: python tools/host_path_bench.py [n_proofs]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402

rsv = rsvload.load_package()
FIX = ["recursive_proof_16_15.bin", "level3-1.bin", "level6-1.bin", "level7-1.bin"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    fx = [np.fromfile(os.path.join(ROOT, "tests", "golden", "proofs", f), dtype=np.uint8) for f in FIX]
    # n distinct host buffers (so that no copy is served from a warm cache line of a shared fixture)
    proofs = [fx[i % 4].copy() for i in range(n)]
    for i in range(5, n, 17):
        proofs[i][100 + i % 1000] ^= 1
    total = sum(p.size for p in proofs)
    import ctypes
    u8p, u64p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint64)
    ctx = rsv.Context(0)
    std = rsv.PcsConfig(20, 5, 8, 16)  # standard_config, examples/multi-proofs/src/main.rs:173-176
    pc = rsv.prepare_cfg(std, n)
    ctx.verify_batch_host(proofs[:256], std)  # warm up: module load, workspace
    ptrs = (ctypes.c_void_p * n)(*[p.ctypes.data for p in proofs])
    lens = np.array([p.size for p in proofs], dtype=np.uint64)
    pi = rsv.make_inputs(rsv.STANDARD_INPUTS)
    acc = np.zeros(n, np.uint8)
    out = {"n": n, "bytes": total}
    for label, env, thr in (("warm_512MB", "512", "8"), ("pipelined_128MB_8t", "128", "8"), ("pipelined_256MB_8t", "256", "8"),
                            ("pipelined_256MB_16t", "256", "16"), ("pipelined_256MB_4t", "256", "4"),
                            ("pipelined_512MB_8t", "512", "8"), ("pipelined_512MB_16t", "512", "16")):
        ctx.set_option("host_chunk_mb", int(env))
        ctx.set_option("host_threads", int(thr))
        acc[:] = 0
        t0 = time.perf_counter()
        rc = rsv.lib.rsv_verify_batch_host(ctx._h, ptrs, lens.ctypes.data_as(u64p), n, pc.ref(), pi, 3, acc.ctypes.data_as(u8p), None)
        dt = time.perf_counter() - t0
        assert rc == 0 and int(acc.sum()) == n - len(range(5, n, 17))
        out[label] = {"s": dt, "proofs_per_s": n / dt, "GBps": total / dt / 1e9}
    # the caller that cooperates: proofs read back to back into the library's pinned arena (rsv_host_alloc) — no gather copy
    arena = rsv.HostArena(total + 4096)
    t0 = time.perf_counter()
    hb = arena.pack(proofs)
    out["arena_fill_s"] = time.perf_counter() - t0   # (the caller's own read / deserialisation, not part of the call)
    lens_a = hb.lens
    for label, mb in (("arena_direct_64MB", 64), ("arena_direct_128MB", 128), ("arena_direct_256MB", 256), ("arena_direct_256MB_again", 256)):
        ctx.set_option("host_chunk_mb", mb)
        acc[:] = 0
        t0 = time.perf_counter()
        rc = rsv.lib.rsv_verify_batch_host(ctx._h, hb.ptrs, lens_a.ctypes.data_as(u64p), n, pc.ref(), pi, 3, acc.ctypes.data_as(u8p), None)
        dt = time.perf_counter() - t0
        assert rc == 0 and int(acc.sum()) == n - len(range(5, n, 17))
        out[label] = {"s": dt, "proofs_per_s": n / dt, "GBps": total / dt / 1e9}
    arena.close()
    blob = np.concatenate(proofs)
    offsets = np.zeros(n + 1, np.uint64)
    offsets[1:] = np.cumsum([p.size for p in proofs], dtype=np.uint64)
    acc = np.zeros(n, np.uint8)
    t0 = time.perf_counter()
    rc = rsv.lib.rsv_verify_batch(blob.ctypes.data_as(u8p), offsets.ctypes.data_as(u64p), n, pc.ref(), pi, 3,
                                  acc.ctypes.data_as(u8p), None, 0)
    dt = time.perf_counter() - t0
    assert rc == 0
    out["one_blob_synchronous"] = {"s": dt, "proofs_per_s": n / dt, "GBps": total / dt / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
