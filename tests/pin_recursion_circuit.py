#!/usr/bin/env python3
"""Pins oracle/recursion_circuit against the reference's fixture chain and records what it finds in
tests/golden/recursion_circuit_pins.json (CPU only, ~10 min; tests/test_recursion_circuit.py re-checks a subset from it).

For every consecutive pair (src, multiplier, dst) of the reference's fixtures — dst is the proof of the circuit that
verifies src `multiplier` times (examples/single-proof/src/main.rs, examples/multi-proofs/src/main.rs:173-295) — the
restated circuit for src is built, padded and turned into the columns the prover committed to: 10 preprocessed + 12 trace
columns of the Plonk component, 40 + 48 of the Poseidon component.  Each column's interpolant, evaluated at dst's OODS
point (from dst's own transcript), must be the sampled value dst carries for that column: 110 equalities in QM31 per
pair.  They hold only if wires, ops, multiplicities, the `variables` vector and the PoseidonFlow are the reference's.

The one thing that has to be searched: the order in which AnswerResults::compute walked its two HashSet<isize> = {0, -1}
(oracle/recursion_circuit/verifier.py) — two bits per copy of the verifier, seeded per process in the reference.  For one
copy the four orders are tried; for five copies the columns' values are affine in the ten bits (a flip moves wire
numbers inside its own copy only), so ten single-flip builds give the differences and the subset is read off."""
import ctypes
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_binding as ob  # noqa: E402
from oracle import recursion_circuit as rc  # noqa: E402
from oracle.recursion_circuit import trace as T  # noqa: E402

C = rc.C

CHAIN = [("small_proof.bin", 1, "recursive_proof_16_15.bin"), ("recursive_proof_16_15.bin", 5, "level1-5.bin"),
         ("level1-5.bin", 1, "level2-1.bin"), ("level2-1.bin", 1, "level3-1.bin"), ("level3-1.bin", 5, "level4-5.bin"),
         ("level4-5.bin", 1, "level5-1.bin"), ("level5-1.bin", 1, "level6-1.bin"), ("level6-1.bin", 1, "level7-1.bin"),
         ("level7-1.bin", 1, "level8-1.bin"), ("level8-1.bin", 1, "level9-1.bin"), ("level9-1.bin", 1, "level10-1.bin"),
         ("level10-1.bin", 1, "level11-1.bin"), ("level11-1.bin", 1, "level12-1.bin"), ("level12-1.bin", 1, "level13-1.bin")]
ORDERS = ((0, -1), (-1, 0))


def round_constants():
    ob.lib.rsvo_round_constants.restype = ctypes.POINTER(ctypes.c_uint32)
    r = [ob.lib.rsvo_round_constants(k) for k in range(3)]
    return ([[int(r[0][16 * a + i]) for i in range(16)] for a in range(4)], [int(r[1][i]) for i in range(14)],
            [[int(r[2][16 * a + i]) for i in range(16)] for a in range(4)])


def read(name):
    with open(os.path.join(ROOT, "tests", "golden", "proofs", name), "rb") as f:
        return f.read()


def target(dst):
    nxt = read(dst)
    tr = ob.transcript_raw(nxt)
    oods = (tuple(int(x) for x in tr[20:24]), tuple(int(x) for x in tr[24:28]))
    w = np.frombuffer(nxt[:8], np.uint32)
    return oods, rc.parse_proof(nxt).sampled_values, int(w[0]), int(w[1])


def check_pair(src, mult, dst, orders, inputs, rcs, evaluators=None):
    """-> (n matching Plonk columns of 22, n matching Poseidon columns of 88, rows, invocations)."""
    oods, want, lp, lq = target(dst)
    c, _, _ = rc.build_circuit(read(src), ob, inputs, mult, shift_order=orders)
    rows, n_flow = c.num_plonk_rows(), len(c.flow)
    assert T.pad(c) == 1 << lp
    pe_p, pe_q = evaluators or (T.PointEvaluator(lp, oods), T.PointEvaluator(lq, oods))
    pre, trace = T.plonk_columns(c)
    ok_p = sum(pe_p.eval(pre[name]) == want[0][k][0] for k, name in enumerate(T.PREPROCESSED))
    ok_p += sum(pe_p.eval(trace[k]) == want[1][k][0] for k in range(12))
    qpre, qtr = T.poseidon_columns(c.flow, rcs, lq, padding_hash=([0] * 8,))
    ok_q = sum(pe_q.eval(qpre[k]) == want[0][10 + k][0] for k in range(40))
    ok_q += sum(pe_q.eval(qtr[k]) == want[1][12 + k][0] for k in range(48))
    return ok_p, ok_q, rows, n_flow


def find_orders(src, mult, dst, inputs):
    oods, want, lp, _ = target(dst)
    pe = T.PointEvaluator(lp, oods)

    goal = [want[0][k][0] for k in range(10)] + [want[1][k][0] for k in range(12)]

    def wires_at(orders):  # all 22 Plonk columns: between them they see every block that moves
        c, _, _ = rc.build_circuit(read(src), ob, inputs, mult, shift_order=orders)
        T.pad(c)
        pre, trace = T.plonk_columns(c)
        return [pe.eval(pre[name]) for name in T.PREPROCESSED] + [pe.eval(trace[k]) for k in range(12)]

    if mult == 1:
        for o in itertools.product(ORDERS, repeat=2):
            if wires_at([o]) == goal:
                return [[list(x) for x in o]]
        raise SystemExit(f"{src}: no order reproduces the Plonk columns")
    base_orders = [[ORDERS[0], ORDERS[0]] for _ in range(mult)]
    base = wires_at([tuple(o) for o in base_orders])
    deltas = []
    for k in range(mult):
        for s in range(2):
            o = [list(x) for x in base_orders]
            o[k][s] = ORDERS[1]
            deltas.append([C.q_sub(x, y) for x, y in zip(wires_at([tuple(x) for x in o]), base)])
            print(f"  flip copy {k} set {s} done", flush=True)
    for bits in itertools.product((0, 1), repeat=2 * mult):
        v = list(base)
        for b, dl in zip(bits, deltas):
            if b:
                v = [C.q_add(x, y) for x, y in zip(v, dl)]
        if v == goal:
            return [[list(ORDERS[bits[2 * k]]), list(ORDERS[bits[2 * k + 1]])] for k in range(mult)]
    raise SystemExit(f"{src}: no combination of orders reproduces the Plonk columns")


def main():
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        man = {e["file"]: e for e in json.load(f)["proofs"]}
    rcs = round_constants()
    out = {"_about": "written by tests/pin_recursion_circuit.py: per fixture pair, the HashSet walk orders under which the restated "
                     "circuit reproduces every sampled value of the next fixture's Plonk (22) and Poseidon (88) columns",
           "pairs": []}
    for src, mult, dst in CHAIN:
        t = time.time()
        inputs = [(i, tuple(v)) for i, v in man[src]["inputs"]]
        orders = find_orders(src, mult, dst, inputs)
        ok_p, ok_q, rows, n_flow = check_pair(src, mult, dst, [tuple(tuple(x) for x in o) for o in orders], inputs, rcs)
        print(f"{src:28s} x{mult} -> {dst:28s} rows {rows:7d} invocations {n_flow:6d} plonk {ok_p}/22 poseidon {ok_q}/88 "
              f"orders {orders}  {time.time() - t:.0f}s", flush=True)
        assert ok_p == 22 and ok_q == 88
        out["pairs"].append({"src": src, "multiplier": mult, "dst": dst, "plonk_rows": rows, "poseidon_invocations": n_flow,
                             "shift_orders": orders})
    with open(os.path.join(ROOT, "tests", "golden", "recursion_circuit_pins.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
