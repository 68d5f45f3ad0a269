#!/usr/bin/env python3
"""Host time of one rsv_verify_batch_dev call (the enqueue: it returns without waiting for the device) against the whole
call + rsv_ctx_synchronize, for small batches, with and without the HIP graph.  gpurun -- 'python tools/enqueue_time.py'"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402

rsv = rsvload.load_package()
import torch  # noqa: E402


def main():
    from tests.conftest import fixture_cfg, read_proof
    pr = read_proof("recursive_proof_16_15.bin")
    cfg = rsv.PreparedCfg([fixture_cfg("recursive_proof_16_15.bin")])
    dev = torch.device("cuda:0")
    for graph in ("off", "on"):
        for n in (1, 16, 128, 1024):
            blob, offsets = rsv.pack([pr] * n)
            d_blob = torch.from_numpy(blob.copy()).to(dev)
            d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
            d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
            ctx = rsv.Context(0)
            ctx.set_option("graph", graph)
            enq, tot = [], []
            for k in range(40):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.verify_batch(d_blob, d_off, n, d_acc, None, cfg)
                t1 = time.perf_counter()
                ctx.synchronize()
                t2 = time.perf_counter()
                if k >= 8:
                    enq.append(t1 - t0)
                    tot.append(t2 - t0)
            assert int(d_acc.sum().item()) == n
            enq.sort(); tot.sort()
            print(f"graph={graph} n={n}: enqueue {enq[len(enq) // 2] * 1e3:.3f} ms, call + synchronize {tot[len(tot) // 2] * 1e3:.3f} ms (min {tot[0] * 1e3:.3f})", flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
