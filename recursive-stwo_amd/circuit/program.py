"""Witness program: the recursion circuit's `variables` vector as a list of instructions a GPU can evaluate for a whole
batch of proofs of one shape (rsv_witness_eval_dev, include/rsv.h).

The reference fills `variables` while it runs the circuit's gadgets on one proof
(constraint_system/src/plonk_with_poseidon.rs:140-283: every add / mul / mul_constant / new_m31 / new_qm31 pushes one
value).  Which gate or hint produces variable k is the same for every proof of a shape, so the gadgets are run ONCE, on a
template proof (circuit/verifier.py), and what they did is written down as one instruction per variable:

    op        a, b, imm                     value of variable dst
    CONST     imm[0..4]                     the constant
    ADD MUL   a, b                          variables[a] + / * variables[b]              (cs.add, cs.mul, the Poseidon gate)
    MULC      a, imm0                       variables[a] * imm0                          (cs.mul_constant)
    COPY      a                             variables[a]                                 (the OODS point re-allocated as a witness)
    INV INV0 QINV CINV   a [, imm0 = part]  inverse hints: M31, M31-or-zero, QM31, one coordinate of a CM31 inverse
    COORD BIT a, imm0                       imm0-th M31 coordinate / bit of variables[a]
    FLOW      imm0 = invocation, imm1 = word      4 words of a PoseidonFlow record (the permutation's outputs are hints)
    WORD WORD4 imm0 = word offset           proof words at a fixed offset (statement, commitments, sampled values)
    FRI_COMMIT LAST_POLY NONCE              proof words behind the variable-length sections (offsets from the parser)
    TRACE_COL imm0 = tree, imm1 = query, imm2 = index       SinglePathMerkleProof::columns (rsv_hints_out::d_trace_cols)
    FRI_COL   imm0 = tree, imm1 = query, imm2 = word        SinglePairMerkleProof self / sibling values (d_fri_cols)

Instructions are sorted by dependency depth ("levels"): everything inside a level only reads variables of earlier levels,
so the GPU runs one launch per level over (instructions of the level) x (proofs).  The hash chains cost no depth — the
outputs of the Poseidon accelerator are hints read from the flow records — which leaves the arithmetic chains (the
composition accumulator, the folds): 265 to 300 levels for 39 000 to 330 000 variables.
"""
from __future__ import annotations

import numpy as np

(CONST, ADD, MUL, MULC, COPY, INV, INV0, QINV, CINV, COORD, BIT, FLOW, WORD, WORD4, FRI_COMMIT, LAST_POLY, NONCE, TRACE_COL,
 FRI_COL) = range(19)
OP_NAMES = ("CONST ADD MUL MULC COPY INV INV0 QINV CINV COORD BIT FLOW WORD WORD4 FRI_COMMIT LAST_POLY NONCE TRACE_COL "
            "FRI_COL").split()
INSTR_WORDS = 8  # op, dst, a, b, imm0..imm3

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


class Program:
    """instr: uint32[n_instr, 8] sorted by level; level_offsets: uint32[n_levels + 1]; n_vars; shape = what a proof must
    look like for this program to apply; flow_wires: uint32[copies * flow_count, 5] = the wire indices of the circuit's
    PoseidonFlow entries (PoseidonEntry::wire of r1..r4, SwapOption::addr; plonk_with_poseidon.rs:117-128), constants of
    the shape that go with the hashes rsv_witness_eval_dev returns in d_flow (invocation k of copy c = row c * flow_count + k)."""

    def __init__(self, instr, level_offsets, n_vars, shape, flow_wires=None):
        self.instr, self.level_offsets, self.n_vars, self.shape = instr, level_offsets, n_vars, shape
        self.flow_wires = flow_wires if flow_wires is not None else np.zeros((shape["copies"] * shape["flow_count"], 5), np.uint32)

    def save(self, path):
        np.savez_compressed(path, instr=self.instr, level_offsets=self.level_offsets, n_vars=np.array([self.n_vars]),
                            shape=np.array([self.shape[k] for k in SHAPE_KEYS], dtype=np.uint32), flow_wires=self.flow_wires)

    RAW_MAGIC = 0x57565352  # "RSVW"

    def save_raw(self, path):
        """The flat file the C++ host mirror loads (host/recursive_stwo.hpp, WitnessProgram::load): "RSVW", version 1, n_vars,
        n_levels, the 9 shape words, level_offsets, instr, flow_wires."""
        head = np.array([self.RAW_MAGIC, 1, self.n_vars, len(self.level_offsets) - 1] + [self.shape[k] for k in SHAPE_KEYS], dtype=np.uint32)
        with open(path, "wb") as f:
            f.write(head.tobytes())
            f.write(np.ascontiguousarray(self.level_offsets, dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(self.instr, dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(self.flow_wires, dtype=np.uint32).tobytes())

    @staticmethod
    def load_raw(path):
        w = np.fromfile(path, dtype=np.uint32)
        if len(w) < 13 or w[0] != Program.RAW_MAGIC or w[1] != 1:
            raise ValueError(f"not a witness program: {path}")
        n_vars, n_levels = int(w[2]), int(w[3])
        shape = dict(zip(SHAPE_KEYS, (int(x) for x in w[4:13])))
        levels = w[13:13 + n_levels + 1].copy()
        at = 13 + n_levels + 1
        n_flow = shape["copies"] * shape["flow_count"]
        if len(w) != at + n_vars * INSTR_WORDS + 5 * n_flow:
            raise ValueError(f"truncated witness program: {path}")
        instr = w[at:at + n_vars * INSTR_WORDS].reshape(n_vars, INSTR_WORDS).copy()
        return Program(instr, levels, n_vars, shape, w[at + n_vars * INSTR_WORDS:].reshape(n_flow, 5).copy())

    @staticmethod
    def load(path):
        z = np.load(path)
        return Program(z["instr"], z["level_offsets"], int(z["n_vars"][0]), dict(zip(SHAPE_KEYS, (int(x) for x in z["shape"]))),
                       z["flow_wires"])


SHAPE_KEYS = ("lp", "lq", "pow_bits", "blowup", "log_last", "nq", "n_inner", "flow_count", "copies")


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
