"""Host side of rsv_witness_eval_dev: the reference's recursion circuit, mirrored so that it can be written down as a
program the GPU evaluates for whole batches (include/rsv.h, "the recursion circuit's witness").

  cs.py        PlonkWithPoseidonConstraintSystem + M31Var / CM31Var / QM31Var     constraint_system/src/plonk_with_poseidon.rs,
                                                                                   primitives/fields/src/*.rs
  gadgets.py   BitsVar, Poseidon2HalfVar, ChannelVar, Poseidon31MerkleHasherVar,   primitives/{bits,poseidon31,channel,merkle,
               circle points, LinePolyVar, query positions                         circle,line,query}/src/lib.rs
  verifier.py  PlonkWithPoseidonProofVar, FiatShamirResults, CompositionCheck,     components/recursive/*/src/lib.rs
               AnswerResults, FoldingResults
  program.py   the instruction list extracted from a run of the above

The gadgets run on ONE template proof per shape, with Python integers, exactly as the reference runs them on every
proof; what they leave behind here is not the witness but the recipe (which gate or hint makes variable k).  The
template's hints — Merkle paths, column values, the outputs of the Poseidon accelerator — come from the GPU's own
verifying pass over the template (rsv_verify_hints), so building a program needs the HIP library like everything else in
this package.  tests/ pin this very code to the reference: fed with the CPU checker's hints instead, the circuit it builds
reproduces the next fixture's sampled values column by column (tests/test_recursion_circuit.py).
"""
from __future__ import annotations

from . import cs, gadgets, program, verifier  # noqa: F401
from .program import Program  # noqa: F401
from .shape import ProofData, PairProof, PathProof, parse_proof  # noqa: F401


def run_circuit(d, permute_or_flow, inputs_list, copies=1, shift_order=((0, -1), (0, -1))):
    """Run the verifier's gadgets `copies` times over the template d (a ProofData with its hint structs filled in) in one
    constraint system, as examples/multi-proofs/src/main.rs:64-141 does.  permute_or_flow: 16 ints -> 16 ints."""
    gadgets.PERMUTE = permute_or_flow
    c = cs.ConstraintSystem()
    orders = shift_order if isinstance(shift_order, list) else [shift_order] * copies
    marks = []
    for k in range(copies):
        pub = [(idx, cs.qm31_constant(c, tuple(int(x) for x in val))) for idx, val in inputs_list]
        marks.append(verifier.verify_in_circuit(c, d, pub, tuple(tuple(o) for o in orders[k])))
    return c, marks


def template_from_gpu(rsv, proof: bytes, cfg, inputs_list, device: int = 0):
    """ProofData of a template proof with the hint structs the circuit consumes, from the GPU's verifying pass.
    Returns (d, flow uint32[count, 32], swap uint8[count])."""
    d = parse_proof(proof)
    count = rsv.poseidon_flow_count(d.lp, d.lq, cfg)
    h = rsv.hints([proof], cfg, d.nq, d.M, d.n_inner, count, inputs_list, device)
    if not h["accept"][0]:
        raise ValueError(f"the template proof does not verify (reason {int(h['reason'][0])})")
    d.fill_hints(h["trace_sib"][0], h["trace_pos"][0], h["trace_cols"][0], h["fri_sib"][0], h["fri_cols"][0])
    return d, h["flow"][0], h["flow_swap"][0]


def build_program(rsv, proof: bytes, cfg, inputs_list=None, copies: int = 1, device: int = 0) -> Program:
    """The witness program of the shape of `proof` (one run of the gadgets over it, ~1 s per 50 000 variables)."""
    inputs_list = rsv.STANDARD_INPUTS if inputs_list is None else inputs_list
    d, flow, swap = template_from_gpu(rsv, proof, cfg, inputs_list, device)
    cursor = [0]

    def from_flow(state):
        # the permutation outputs are the GPU's own PoseidonFlow records, consumed in invocation order; the inputs must
        # be the ones the circuit is about to permute, or the record order is not the circuit's
        k = cursor[0] % len(flow)
        rec = flow[k]
        given = [int(x) for x in (list(rec[8:16]) + list(rec[0:8]) if swap[k] else rec[0:16])]
        if given != [int(x) for x in state]:
            raise ValueError(f"PoseidonFlow record {k} is not the circuit's invocation {cursor[0]}")
        cursor[0] += 1
        return [int(x) for x in rec[16:32]]

    c, _ = run_circuit(d, from_flow, inputs_list, copies)
    if cursor[0] != copies * len(flow):
        raise ValueError("the circuit made another number of Poseidon invocations than the flow holds")
    return program.extract(c, d, copies)
