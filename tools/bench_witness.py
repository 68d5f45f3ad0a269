#!/usr/bin/env python3
"""Throughput of rsv_witness_eval_dev (the recursion circuit's `variables` for a batch), one JSON line.

    python tools/bench_witness.py --fixture level10-1.bin --proofs 1024 --steps 5 [--copies 1]

The batch is `--proofs` copies of the fixture, every 17th with one flipped bit (rejected: its row is not evaluated into
anything meaningful, the cost is the same).  Timed: the whole call (verifying pass with the hint outputs the witness
needs + level launches + transpose) and, for the split, the verifying pass alone with the same hint outputs.
roofline: the level kernels are HBM streaming — per instruction and proof 16 B written, 16 B per variable operand read,
4 / 16 B of hint source; the transpose reads and writes 16 B per variable — so algorithmic bytes per proof follow from
the program; achieved = those bytes x proofs / (whole call - verifying pass)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="level10-1.bin")
    ap.add_argument("--proofs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--copies", type=int, default=1)
    ap.add_argument("--no-split", action="store_true", help="skip the verifying-pass-only timing (for a kernel trace whose last step is a witness call)")
    ap.add_argument("--small-max", type=int, default=0, help="RSV_OPT_WITNESS_SMALL_MAX (0 = the library's default)")
    ap.add_argument("--small-log", type=int, default=0, help="RSV_OPT_WITNESS_SMALL_LOG (0 = the library's default)")
    ap.add_argument("--layout", choices=["by_proof", "by_variable"], default="by_proof",
                    help="by_variable: d_variables[variable][proof] as the level kernels write it (RSV_OPT_WITNESS_LAYOUT = 2): no transpose")
    args = ap.parse_args()
    import rsvload
    rsv = rsvload.load_package()
    import torch
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        man = {e["file"]: e for e in json.load(f)["proofs"]}
    e = man[args.fixture]
    inputs = [(i, tuple(v)) for i, v in e["inputs"]]
    cfg = rsv.PcsConfig(e["pow_bits"], e["log_blowup_factor"], e["log_last_layer_degree_bound"], e["n_queries"])
    proof = bench.read_fixture(args.fixture)
    t0 = time.perf_counter()
    wp = rsv.WitnessProgram.build(proof, cfg, inputs, copies=args.copies)
    build_s = time.perf_counter() - t0
    prog = wp.export()
    P_ = rsv.witness_program
    ops = prog.instr[:, 0]
    n_two = int(np.isin(ops, (P_.ADD, P_.MUL)).sum())
    n_one = int(np.isin(ops, (P_.MULC, P_.COPY, P_.INV, P_.INV0, P_.QINV, P_.CINV, P_.COORD, P_.BIT)).sum())
    n_h16 = int(np.isin(ops, (P_.FLOW, P_.WORD4, P_.FRI_COMMIT, P_.LAST_POLY, P_.FRI_COL)).sum())
    n_h4 = int(np.isin(ops, (P_.WORD, P_.NONCE, P_.TRACE_COL)).sum())
    bytes_per_proof = 16 * prog.n_vars + 32 * n_two + 16 * n_one + 16 * n_h16 + 4 * n_h4 + (32 * prog.n_vars if args.layout == "by_proof" else 0)
    n = args.proofs
    dev = torch.device("cuda:0")
    batch = [proof] * n
    tampered = list(range(5, n, 17))
    for i in tampered:
        b = bytearray(proof)
        b[4000 + (i * 7919) % (len(proof) - 8000)] ^= 1
        batch[i] = bytes(b)
    blob, offsets = rsv.pack(batch)
    d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_vars = torch.empty((n, prog.n_vars, 4), dtype=torch.int32, device=dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx = rsv.Context(0)
    ctx.set_option("witness_layout", args.layout)
    ctx.set_option("witness_small_max", args.small_max)
    ctx.set_option("witness_small_log", args.small_log)
    s = prog.shape
    M = max(s["lp"] + 1, s["lq"] + 2) + s["blowup"]
    hint = dict(shape=(s["nq"], M, s["n_inner"]),  # what the witness call asks the verifying pass for: column values + the flow
                d_trace_cols=torch.empty((n, 4, s["nq"], 64), dtype=torch.int32, device=dev),
                d_fri_cols=torch.empty((n, 1 + s["n_inner"], s["nq"], 3, 8), dtype=torch.int32, device=dev),
                d_flow=torch.empty((n, s["flow_count"], 32), dtype=torch.int32, device=dev),
                d_flow_swap=torch.empty((n, s["flow_count"]), dtype=torch.uint8, device=dev))

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t) / args.steps * 1e3

    whole_ms = timed(lambda: ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=inputs))
    acc = d_acc.cpu().numpy()
    want = np.ones(n, np.uint8)
    want[tampered] = 0
    if not np.array_equal(acc, want):
        raise SystemExit("verdict mismatch")
    hints_ms = 0.0 if args.no_split else timed(lambda: ctx.verify_hints(d_blob, d_off, n, d_acc, None, cfg, inputs, **hint))
    eval_ms = max(whole_ms - hints_ms, 1e-6)
    achieved = bytes_per_proof * n / (eval_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "recursion_circuit_witnesses_per_s", "value": n / (whole_ms * 1e-3), "unit": "proofs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": whole_ms, "higher_is_better": True, "dtype": "u32 (M31 / QM31)", "data": "synthetic",
        "config": {"workload": f"witness of the circuit verifying {args.fixture} x{args.copies}", "proofs": n, "variables_per_proof": prog.n_vars,
                   "levels": int(len(prog.level_offsets) - 1), "poseidon_invocations": s["flow_count"] * args.copies, "layout": args.layout,
                   "output_bytes_per_proof": 16 * prog.n_vars, "scratch_bytes": wp.scratch_bytes(n), "program_build_s": round(build_s, 2)},
        "split_ms": {"verifying_pass_with_hints": hints_ms, "levels_and_transpose": eval_ms},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                     "note": f"the program ({len(prog.level_offsets) - 1} levels: one k_witness_small launch for batches up to 1 024 proofs of a short program, else k_witness_level[_wide] for the wide head, a launch per level, and every level behind it in one k_witness_small launch) + k_witness_transpose; algorithmic bytes/proof "
                             f"{bytes_per_proof} = 16 B written per variable, 16 B per variable operand ({2 * n_two + n_one}), hint sources, "
                             "32 B per variable for the transpose"},
        "kernel_sources_sha": bench.kernel_sources_sha()}))


if __name__ == "__main__":
    main()
