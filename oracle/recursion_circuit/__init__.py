"""TEST INFRASTRUCTURE (oracle): the checker of the recursion circuit.

The circuit's gadgets themselves (constraint system, field variables, Poseidon / Merkle / channel / circle gadgets, the
five verifier stages) are host-side PRODUCT code: recursive-stwo_amd/circuit, which runs them once per proof shape to
write the witness program the GPU evaluates.  What lives here is what holds that code to the reference:
  inputs.py   feeds the gadgets with the CPU oracle's hints (oracle/rsv_oracle.c) instead of the GPU's, and with the C
              oracle's Poseidon2 permutation;
  trace.py    pads the resulting constraint system as the reference does, builds the columns the reference's prover
              commits to (Plonk: 10 preprocessed + 12 trace; Poseidon: 40 + 48) and evaluates their interpolants at a
              point — so that they can be compared with the `sampled_values` of the NEXT proof of the reference's fixture
              chain, which is the proof of exactly this circuit (tests/test_recursion_circuit.py,
              tools/pin_recursion_circuit.py).
Values computed by running the gadgets here are also what the GPU's evaluation of the program is compared with
(tests/test_gpu_parity.py).  Only tests/ and the parity tooling import this package."""
import numpy as np

import rsvload

circuit = rsvload.load_package().circuit
C, gadgets, program, parse_proof = circuit.cs, circuit.gadgets, circuit.program, circuit.parse_proof

from . import inputs, trace  # noqa: E402,F401


def build_circuit(proof: bytes, ob, inputs_list=None, multipliers=1, shift_order=((0, -1), (0, -1))):
    """The constraint system the reference leaves after verifying `proof` `multipliers` times in one circuit
    (examples/multi-proofs/src/main.rs:64-141), before cs.pad().  `ob` = tests/oracle_binding.  shift_order: one pair of
    orders (circuit/verifier.py) for every copy, or a list with one pair per copy.  -> (cs, ProofData, marks)."""

    def permute(state):
        return ob.poseidon2_permute(np.array(state, dtype=np.uint32))[0].tolist()

    d = inputs.build_inputs(proof, ob, inputs_list)
    c, marks = circuit.run_circuit(d, permute, ob.STANDARD_INPUTS if inputs_list is None else inputs_list, multipliers, shift_order)
    return c, d, marks
