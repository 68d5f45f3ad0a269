"""Mutant generators shared by the sanitizer run of the oracle (CPU) and the GPU parity fuzz: structural mutants,
every length prefix perturbed, random byte corruption, truncations."""
import numpy as np

from tests import oracle_binding as ob


def mutants_of(proof: bytes, rng, n_random: int):
    batch = [b for _, b in ob.structural_mutants(proof)] + [b for _, b in ob.noncanonical_structural_mutants(proof)]
    for pos, n, _ in ob.proof_layout(proof)["prefixes"]:
        for val in {max(n - 1, 0), n + 1, 0, 0xFFFFFFFF, (1 << 32) + n, 8 * n + 3} - {n}:
            b = bytearray(proof)
            b[4 * pos:4 * pos + 8] = int(val).to_bytes(8, "little")
            batch.append(bytes(b))
    # non-canonical words: the word right behind every length prefix (a field element or a hash word wherever the
    # sequence is not empty: both sides must say PARSE) and the proof's final word, last_layer_poly.log_size, which the
    # reference never reads (any value verifies)
    words = np.frombuffer(proof, np.uint32)
    for pos, n, _ in ob.proof_layout(proof)["prefixes"]:
        if n and pos + 2 < len(words):
            for val in (0x7FFFFFFF, 0xFFFFFFFF):
                w = words.copy()
                w[pos + 2] = val
                batch.append(w.tobytes())
    for val in (0, 9, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF):
        w = words.copy()
        w[-1] = val
        batch.append(w.tobytes())
    for k in range(n_random):
        b = bytearray(proof)
        for _ in range(1 + k % 5):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        batch.append(bytes(b))
    for cut in (0, 4, 60, 64, 3580, 3584, (len(proof) // 8) * 4, len(proof) - 4):
        batch.append(proof[:cut])
    return batch


if __name__ == "__main__":
    # python -m tests.mutants <fixture>...: run the oracle over the mutants (used under LD_PRELOAD=libasan by
    # tests/test_oracle.py::test_oracle_is_clean_under_sanitizers)
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(5)
    total = 0
    for name in sys.argv[1:]:
        with open(os.path.join(root, "tests", "golden", "proofs", name), "rb") as f:
            proof = f.read()
        batch = mutants_of(proof, rng, 40) + [proof]
        acc, reason = ob.verify_batch(batch, ob.header_cfg(proof))  # the genuine fixture's header == its manifest configuration
        total += len(batch)
    print("sanitized oracle ran", total, "proofs with", os.path.basename(ob.lib._name))
