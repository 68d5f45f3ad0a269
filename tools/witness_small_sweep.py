"""Sweep of the one-launch witness form (RSV_OPT_WITNESS_SMALL_MAX / _LOG) against the level-per-launch form: time of the
levels (+ transpose) for a batch, and bit-equality of the outputs.  Usage: python tools/witness_small_sweep.py [fixture]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402

rsv = rsvload.load_package()
import json  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "level10-1.bin"
    proof = open(os.path.join(ROOT, "tests", "golden", "proofs", name), "rb").read()
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        e = {e["file"]: e for e in json.load(f)["proofs"]}[name]
    inputs = [(i, tuple(v)) for i, v in e["inputs"]]
    cfg = rsv.PcsConfig(e["pow_bits"], e["log_blowup_factor"], e["log_last_layer_degree_bound"], e["n_queries"])
    wp = rsv.WitnessProgram.build(proof, cfg, inputs)
    dev = torch.device("cuda:0")
    ctx = rsv.Context(0)
    for n in (1, 2, 4, 8, 64, 256, 1024, 2048):
        blob, offsets = rsv.pack([proof] * n)
        d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
        d_vars = torch.zeros((n, wp.n_vars, 4), dtype=torch.int32, device=dev)
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)

        def run(small_max, small_log, steps=4):
            ctx.set_option("witness_small_max", small_max)
            ctx.set_option("witness_small_log", small_log)
            d_vars.zero_()
            ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=inputs)
            ctx.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=inputs)
            ctx.synchronize()
            return (time.perf_counter() - t) / steps * 1e3, d_vars.clone()

        base_ms, base = run(1, 0)
        row = [f"n {n:5d}  levels {base_ms:7.3f} ms |"]
        ms, got = run(0, 0)
        row.append(f" default: {ms:7.3f}{'' if torch.equal(got, base) else ' MISMATCH'}")
        for lg in (2, 3):
            ms, got = run(n + 1, lg + 1)
            ok = bool(torch.equal(got, base))
            row.append(f" one launch, {1 << lg if n > 4 else 'n'} per workgroup: {ms:7.3f}{'' if ok else ' MISMATCH'}")
        print("".join(row), flush=True)
        del d_blob, d_vars, base
    # the level form with its tail in one launch (RSV_OPT_WITNESS_WALK_LOG), mid-size batches
    ctx.set_option("witness_small_max", 1)
    for n in ([1024, 2048, 4096] if wp.n_vars > 200000 else [2048, 4096, 8192, 16384]):
        blob, offsets = rsv.pack([proof] * n)
        d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
        d_vars = torch.zeros((n, wp.n_vars, 4), dtype=torch.int32, device=dev)
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)

        def run2(walk_log, steps=4):
            ctx.set_option("witness_walk_log", walk_log)
            d_vars.zero_()
            ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=inputs)
            ctx.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=inputs)
            ctx.synchronize()
            return (time.perf_counter() - t) / steps * 1e3

        base_ms = run2(1)
        base = d_vars.cpu()
        row = [f"n {n:5d}  levels {base_ms:7.3f} ms |"]
        ms = run2(0)
        row.append(f" auto: {ms:7.3f}{'' if torch.equal(d_vars.cpu(), base) else ' MISMATCH'}")
        for lg in (3, 4, 5):
            ms = run2(lg + 1)
            ok = bool(torch.equal(d_vars.cpu(), base))
            row.append(f" tail in one launch, {1 << lg} per workgroup: {ms:7.3f}{'' if ok else ' MISMATCH'}")
        print("".join(row), flush=True)
        del d_blob, d_vars, base
    ctx.close()
    wp.close()


if __name__ == "__main__":
    main()
