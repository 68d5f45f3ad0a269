"""TEST INFRASTRUCTURE (oracle): the witness program as the CPU restatement of the circuit derives it (extract) and a
host interpreter of programs in Python integers (interpret) — what the library's own builder
(rsv_witness_program_build, csrc/circuit_*.hpp) and the GPU's evaluation (k_witness.hpp) are compared with.
The instruction table: recursive-stwo_amd/witness_program.py."""
from __future__ import annotations

import numpy as np

(CONST, ADD, MUL, MULC, COPY, INV, INV0, QINV, CINV, COORD, BIT, FLOW, WORD, WORD4, FRI_COMMIT, LAST_POLY, NONCE, TRACE_COL,
 FRI_COL) = range(19)
OP_NAMES = ("CONST ADD MUL MULC COPY INV INV0 QINV CINV COORD BIT FLOW WORD WORD4 FRI_COMMIT LAST_POLY NONCE TRACE_COL "
            "FRI_COL").split()
INSTR_WORDS = 8  # op, dst, a, b, imm0..imm3
SHAPE_KEYS = ("lp", "lq", "pow_bits", "blowup", "log_last", "nq", "n_inner", "flow_count", "copies")


class Program:
    """instr uint32[n_vars, 8] sorted by level, level_offsets uint32[n_levels + 1], shape, flow_wires uint32[copies *
    flow_count, 5] — the same fields as the product's container (recursive-stwo_amd/witness_program.py), nothing shared."""

    def __init__(self, instr, level_offsets, n_vars, shape, flow_wires):
        self.instr, self.level_offsets, self.n_vars, self.shape, self.flow_wires = instr, level_offsets, n_vars, shape, flow_wires

W_LP, W_LQ, W_PLONK_SUM, W_POSEIDON_SUM, W_COMMIT0 = 0, 1, 2, 6, 17


def sample_offsets():
    """Word offset of every sampled value (tree, column, sample) — csrc/layout.hpp make_sample_table."""
    plonk, poseidon = (10, 12, 8), (40, 48, 8)
    off, pos = {}, 49 + 2
    for t in range(4):
        pos += 2
        n_cols = 8 if t == 3 else plonk[t] + poseidon[t]
        for c in range(n_cols):
            pos += 2
            for s in range(2 if (t == 2 and (c & 4)) else 1):
                off[(t, c, s)] = pos
                pos += 4
    return off


def _hint_instr(tag, d, samples):
    """tag (set by the gadget that allocated the witness) -> (op, a, b, imm0..3)."""
    kind = tag[0]
    if kind == "in":
        what = tag[1]
        if what == "lp":
            return (WORD, 0, 0, W_LP, 0, 0, 0)
        if what == "lq":
            return (WORD, 0, 0, W_LQ, 0, 0, 0)
        if what == "plonk_sum":
            return (WORD4, 0, 0, W_PLONK_SUM, 0, 0, 0)
        if what == "poseidon_sum":
            return (WORD4, 0, 0, W_POSEIDON_SUM, 0, 0, 0)
        if what == "commit":
            return (WORD4, 0, 0, W_COMMIT0 + 8 * tag[2] + 4 * tag[3], 0, 0, 0)
        if what == "sample":
            return (WORD4, 0, 0, samples[(tag[2], tag[3], tag[4])], 0, 0, 0)
        if what == "fri_commit":
            return (FRI_COMMIT, 0, 0, tag[2], tag[3], 0, 0)
        if what == "last_poly":
            return (LAST_POLY, 0, 0, tag[2], 0, 0, 0)
        if what == "nonce":
            return (NONCE, 0, 0, tag[2], 0, 0, 0)
    if kind == "inv":
        return (INV, tag[1], 0, 0, 0, 0, 0)
    if kind == "inv_or_zero":
        return (INV0, tag[1], 0, 0, 0, 0, 0)
    if kind == "qinv":
        return (QINV, tag[1], 0, 0, 0, 0, 0)
    if kind == "cinv":
        return (CINV, tag[1], 0, tag[2], 0, 0, 0)
    if kind == "coord":
        return (COORD, tag[1], 0, tag[2], 0, 0, 0)
    if kind == "bit":
        return (BIT, tag[1], 0, tag[2], 0, 0, 0)
    if kind == "perm":  # (flow index, output half, left/right QM31 of that half)
        return (FLOW, 0, 0, tag[1], 16 + 8 * tag[2] + 4 * tag[3], 0, 0)
    if kind == "copy":
        return (COPY, tag[1], 0, 0, 0, 0, 0)
    if kind == "path":  # ("path", tree, query, "col", log_size, j)
        t, i, log_size, j = tag[1], tag[2], tag[4], tag[5]
        return (TRACE_COL, 0, 0, t, i, d.trace_col_base[(t, log_size)] + j, 0)
    if kind == "pair":  # ("pair", tree, query, "self" | "sib", log_size)
        tree, i, which, log_size = tag[1], tag[2], tag[3], tag[4]
        c = d.fri_col_level[(tree, log_size)]
        return (FRI_COL, 0, 0, tree, i, 8 * c + (0 if which == "self" else 4), 0)
    raise ValueError(f"witness of unknown provenance: {tag}")


def extract(cs, d, copies=1) -> Program:
    """cs: the ConstraintSystem after the gadgets ran on the template proof; d: the template's ProofData."""
    n_vars = len(cs.variables)
    samples = sample_offsets()
    flow_per_copy = len(cs.flow) // copies
    rows = np.zeros((n_vars, INSTR_WORDS), dtype=np.int64)
    depth = np.zeros(n_vars, dtype=np.int64)
    for k, origin in enumerate(cs.origin):
        kind = origin[0]
        if kind == "const":
            v = cs.variables[k]
            ins = (CONST, 0, 0, v[0], v[1], v[2], v[3])
        elif kind == "add":
            ins = (ADD, origin[1], origin[2], 0, 0, 0, 0)
        elif kind == "mul":
            ins = (MUL, origin[1], origin[2], 0, 0, 0, 0)
        elif kind == "mulc":
            ins = (MULC, origin[1], 0, origin[2], 0, 0, 0)
        else:
            if origin[1] is None:
                raise ValueError(f"variable {k}: witness without provenance")
            ins = _hint_instr(origin[1], d, samples)
            if ins[0] == FLOW:  # every copy of the verifier reads the same proof's flow
                ins = (FLOW, 0, 0, ins[3] % flow_per_copy, ins[4], 0, 0)
        rows[k] = (ins[0], k) + tuple(ins[1:])
        op = ins[0]
        if op in (ADD, MUL):
            depth[k] = 1 + max(depth[ins[1]], depth[ins[2]])
        elif op in (MULC, COPY, INV, INV0, QINV, CINV, COORD, BIT):
            depth[k] = 1 + depth[ins[1]]
    order = np.argsort(depth, kind="stable")
    instr = rows[order].astype(np.uint32)
    n_levels = int(depth.max()) + 1
    level_offsets = np.searchsorted(depth[order], np.arange(n_levels + 1)).astype(np.uint32)
    shape = {"lp": d.lp, "lq": d.lq, "pow_bits": d.pow_bits, "blowup": d.blowup, "log_last": d.log_last, "nq": d.nq,
             "n_inner": d.n_inner, "flow_count": flow_per_copy, "copies": copies}
    flow_wires = np.array([[e1[0], e2[0], e3[0], e4[0], addr] for (e1, e2, e3, e4, addr, _sw) in cs.flow], dtype=np.uint32)
    return Program(instr, level_offsets, n_vars, shape, flow_wires)


def interpret(program: Program, sources):
    """Reference interpreter of a program in Python integers (host logic for tests of the program itself — the product
    evaluates programs on the GPU).  sources: object with word(i), fri_commit(layer, half), last_poly(k), nonce(part),
    flow(invocation, word), trace_col(t, i, j), fri_col(tree, i, word) returning ints / 4-tuples."""
    from . import cs as C
    P = C.P
    v = [None] * program.n_vars
    for op, dst, a, b, i0, i1, i2, i3 in program.instr.tolist():
        if op == CONST:
            r = (i0, i1, i2, i3)
        elif op == ADD:
            r = C.q_add(v[a], v[b])
        elif op == MUL:
            r = C.q_mul(v[a], v[b])
        elif op == MULC:
            r = C.q_scale(v[a], i0)
        elif op == COPY:
            r = v[a]
        elif op == INV:
            r = (C.m_inv(v[a][0]), 0, 0, 0)
        elif op == INV0:
            r = (C.m_inv(v[a][0]) if v[a][0] else 0, 0, 0, 0)
        elif op == QINV:
            r = C.q_inv(v[a])
        elif op == CINV:
            r = (C.c_inv((v[a][0], v[a][1]))[i0], 0, 0, 0)
        elif op == COORD:
            r = (v[a][i0], 0, 0, 0)
        elif op == BIT:
            r = ((v[a][0] >> i0) & 1, 0, 0, 0)
        elif op == FLOW:
            r = sources.flow(i0, i1)
        elif op == WORD:
            r = (sources.word(i0), 0, 0, 0)
        elif op == WORD4:
            r = tuple(sources.word(i0 + k) for k in range(4))
        elif op == FRI_COMMIT:
            r = sources.fri_commit(i0, i1)
        elif op == LAST_POLY:
            r = sources.last_poly(i0)
        elif op == NONCE:
            r = (sources.nonce(i0), 0, 0, 0)
        elif op == TRACE_COL:
            r = (sources.trace_col(i0, i1, i2), 0, 0, 0)
        elif op == FRI_COL:
            r = sources.fri_col(i0, i1, i2)
        else:
            raise ValueError(op)
        v[dst] = tuple(int(x) % P for x in r)
    return v
