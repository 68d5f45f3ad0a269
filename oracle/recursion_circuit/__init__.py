"""TEST INFRASTRUCTURE (oracle): the reference's recursion circuit restated in Python — the checker of the witness path.

  cs.py, gadgets.py, verifier.py   the constraint system, the gadgets and the five verifier stages, run with Python integers on
                                   one proof (hints from the CPU oracle, oracle/rsv_oracle.c): `variables`, the gate lists, the
                                   PoseidonFlow with its wires — what the reference's ConstraintSystemRef holds before pad()
  program.py                       the witness program derived from such a run + a host interpreter of programs
  trace.py                         pad / multiplicities / the committed columns (Plonk 10 + 12, Poseidon 40 + 48) and the
                                   value of a column's interpolant at a point
What pins it to the reference: every fixture is the proof of the circuit that verifies the previous one, so the columns of
the circuit restated for fixture K, evaluated at fixture K+1's OODS point, must be K+1's sampled values — and are, all 110,
for all 14 consecutive pairs (tests/test_recursion_circuit.py, tests/pin_recursion_circuit.py).
What it checks: the library's own builder (C++, rsv_witness_program_build) must produce this package's program byte for
byte, and the GPU's evaluation must produce this package's `variables` (tests/test_witness_gpu.py).
Only tests/ and the parity tooling import this package."""
import numpy as np

from . import cs, gadgets, inputs, program, shape, trace, verifier  # noqa: F401
from .shape import parse_proof  # noqa: F401

C = cs


def run_circuit(d, permute, inputs_list, copies=1, shift_order=((0, -1), (0, -1))):
    """Run the verifier's gadgets `copies` times over d (a ProofData with its hint structs filled in) in one constraint
    system, as examples/multi-proofs/src/main.rs:64-141 does.  permute: 16 ints -> 16 ints."""
    gadgets.PERMUTE = permute
    c = cs.ConstraintSystem()
    orders = shift_order if isinstance(shift_order, list) else [shift_order] * copies
    marks = []
    for k in range(copies):
        pub = [(idx, cs.qm31_constant(c, tuple(int(x) for x in val))) for idx, val in inputs_list]
        marks.append(verifier.verify_in_circuit(c, d, pub, tuple(tuple(o) for o in orders[k])))
    return c, marks


def build_circuit(proof: bytes, ob, inputs_list=None, multipliers=1, shift_order=((0, -1), (0, -1))):
    """The constraint system the reference leaves after verifying `proof` `multipliers` times in one circuit, before
    cs.pad().  `ob` = tests/oracle_binding.  shift_order: one pair of orders (verifier.py) for every copy, or a list with one
    pair per copy.  -> (cs, ProofData, marks)."""

    def permute(state):
        return ob.poseidon2_permute(np.array(state, dtype=np.uint32))[0].tolist()

    d = inputs.build_inputs(proof, ob, inputs_list)
    c, marks = run_circuit(d, permute, ob.STANDARD_INPUTS if inputs_list is None else inputs_list, multipliers, shift_order)
    return c, d, marks
