"""Generates the reject fixtures that sit BEHIND the proof-of-work check (committed next to this script):

  small_proof_composition.bin  small_proof.bin with one sampled value changed and the nonce re-ground: passes PoW and
                               the logup check, fails the OODS composition identity (RSV_R_COMPOSITION)
  small_proof_dup_query.bin    small_proof.bin with a re-ground nonce whose 16 query positions contain a duplicate
                               (RSV_R_DUP_QUERY; the reference asserts, components/recursive/answer/src/lib.rs:190-195)

  recursive_proof_16_15_composition.bin   the same for a big shape (lp 16 / lq 15, 233-step transcript)
  level1-5_dup_query.bin       level1-5.bin (80 queries at log 21) re-ground to a nonce whose positions collide: the
                               duplicate path of the 80-lane plan kernel

All are derived from the reference's fixtures with the oracle's rsvo_grind_nonce (a brute-force search of ~2^20 and
~2^28 channel permutations).  Usage: python tests/golden/make_reject_fixtures.py   (the second search runs 8 processes
for a couple of minutes)."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_binding as ob  # noqa: E402

HERE = os.path.join(ROOT, "tests", "golden", "proofs")


CFG = {"small_proof.bin": (20, 5, 2, 16), "recursive_proof_16_15.bin": (20, 5, 8, 16), "level1-5.bin": (20, 1, 8, 80)}
STD = [(1, (1, 0, 0, 0)), (2, (0, 1, 0, 0)), (3, (0, 0, 1, 0))]


def _search(args):
    proof, start = args
    try:
        return ob.grind_nonce(proof, want_duplicate_query=True, start=start, max_tries=1 << 29)
    except RuntimeError:
        return None


def main():
    proof = open(os.path.join(HERE, "small_proof.bin"), "rb").read()
    w = np.frombuffer(proof, dtype=np.uint32).copy()
    w[57] = (int(w[57]) + 1) % 0x7FFFFFFF  # first word of the first sampled value (samples start at word 51)
    comp = ob.grind_nonce(w.tobytes())
    open(os.path.join(HERE, "small_proof_composition.bin"), "wb").write(comp)
    print("composition fixture:", ob.verify_batch([comp], ob.PcsConfig(20, 5, 2, 16), [(1, (1, 0, 0, 0))]))
    with mp.Pool(8) as pool:
        for res in pool.imap_unordered(_search, [(proof, k << 40) for k in range(1, 65)]):
            if res is not None:
                open(os.path.join(HERE, "small_proof_dup_query.bin"), "wb").write(res)
                print("dup-query fixture:", ob.verify_batch([res], ob.PcsConfig(20, 5, 2, 16), [(1, (1, 0, 0, 0))]))
                pool.terminate()
                break


def big_shapes():
    proof = open(os.path.join(HERE, "recursive_proof_16_15.bin"), "rb").read()
    w = np.frombuffer(proof, dtype=np.uint32).copy()
    w[57] = (int(w[57]) + 1) % 0x7FFFFFFF
    comp = ob.grind_nonce(w.tobytes())
    open(os.path.join(HERE, "recursive_proof_16_15_composition.bin"), "wb").write(comp)
    print("16_15 composition fixture:", ob.verify_batch([comp], ob.PcsConfig(*CFG["recursive_proof_16_15.bin"]), STD))
    proof = open(os.path.join(HERE, "level1-5.bin"), "rb").read()
    with mp.Pool(8) as pool:
        for res in pool.imap_unordered(_search, [(proof, k << 40) for k in range(1, 65)]):
            if res is not None:
                open(os.path.join(HERE, "level1-5_dup_query.bin"), "wb").write(res)
                print("level1-5 dup-query fixture:", ob.verify_batch([res], ob.PcsConfig(*CFG["level1-5.bin"]), STD))
                pool.terminate()
                break


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big_shapes()
    else:
        main()
