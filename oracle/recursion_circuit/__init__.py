"""TEST INFRASTRUCTURE (oracle): the reference's recursion circuit restated (cs.py, gadgets.py, verifier.py, inputs.py).
Only tests/ and the parity tooling import this package."""
from . import cs, gadgets, inputs, verifier  # noqa: F401


def build_circuit(proof: bytes, ob, inputs_list=None, multipliers=1, shift_order=((0, -1), (0, -1))):
    """The constraint system the reference leaves after verifying `proof` `multipliers` times in one circuit
    (examples/multi-proofs/src/main.rs:64-141), before cs.pad().  `ob` = tests/oracle_binding."""
    import numpy as np

    def permute(state):
        return ob.poseidon2_permute(np.array(state, dtype=np.uint32))[0].tolist()

    gadgets.PERMUTE = permute
    d = inputs.build_inputs(proof, ob, inputs_list)
    c = cs.ConstraintSystem()
    marks = []
    for _ in range(multipliers):
        pub = []
        for idx, val in (ob.STANDARD_INPUTS if inputs_list is None else inputs_list):
            pub.append((idx, cs.qm31_constant(c, tuple(int(x) for x in val))))
        marks.append(verifier.verify_in_circuit(c, d, pub, shift_order))
    return c, d, marks
