"""Witness program container (host side of rsv_witness_program_*, include/rsv.h): the recursion circuit's `variables`
vector as a list of instructions the GPU evaluates for a whole batch of proofs of one shape.

The reference fills `variables` while it runs the circuit's gadgets on one proof
(constraint_system/src/plonk_with_poseidon.rs:140-283: every add / mul / mul_constant / new_m31 / new_qm31 pushes one
value).  Which gate or hint produces variable k is the same for every proof of a shape, so the library runs its mirror of
the gadgets ONCE, on a template proof (rsv_witness_program_build, csrc/circuit_*.hpp), and writes down one instruction per
variable:

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
so the GPU runs one launch per level over (instructions of the level) x (proofs), and the narrow tail in one launch.  The hash chains cost no depth — the
outputs of the Poseidon accelerator are hints read from the flow records — which leaves the arithmetic chains (the
composition accumulator, the folds): 265 levels for 39 000 to 340 000 variables.
"""
from __future__ import annotations

import numpy as np

(CONST, ADD, MUL, MULC, COPY, INV, INV0, QINV, CINV, COORD, BIT, FLOW, WORD, WORD4, FRI_COMMIT, LAST_POLY, NONCE, TRACE_COL,
 FRI_COL) = range(19)
OP_NAMES = ("CONST ADD MUL MULC COPY INV INV0 QINV CINV COORD BIT FLOW WORD WORD4 FRI_COMMIT LAST_POLY NONCE TRACE_COL "
            "FRI_COL").split()
INSTR_WORDS = 8  # op, dst, a, b, imm0..imm3
SHAPE_KEYS = ("lp", "lq", "pow_bits", "blowup", "log_last", "nq", "n_inner", "flow_count", "copies")


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
