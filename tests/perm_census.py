"""How many Poseidon2 permutations does each kernel execute per proof, against the oracle's count of the batched
minimum?  Needs the diagnostic build: make -C recursive-stwo_amd/csrc count.  Usage: python tests/perm_census.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402

rsv = rsvload.load_package(lib_path=os.path.join(ROOT, "recursive-stwo_amd", "csrc", "librsv_hip_count.so"))


TAGS = {0: "other", 1: "k_transcript", 2: "k_row_hash", 3: "k_trace_merkle", 4: "k_pair_merkle"}


def counter():
    out = (ctypes.c_ulonglong * 16)()
    assert rsv.lib.rsv_debug_perm_counter(out) == 0
    return [(int(out[2 * t]), int(out[2 * t + 1])) for t in range(8)]


def main():
    n = 4096
    for name in ("recursive_proof_16_15.bin", "level3-1.bin", "level1-5.bin", "level12-1.bin"):
        proof = open(os.path.join(ROOT, "tests", "golden", "proofs", name), "rb").read()
        want = ob.perm_count(proof)
        counter()
        for env in ("1", "0"):
            rsv.set_default_option("tree_cap", "on" if env == "1" else "off")
            rsv.set_default_option("transcript_form", "lane")
            acc, _ = rsv.verify_batch([proof] * n, ob.header_cfg(proof))  # genuine fixture: header == manifest configuration
            assert acc.all()
            per = counter()
            lanes, waves = sum(a for a, _ in per), sum(b for _, b in per)
            print("   " + ", ".join(f"{TAGS[t]} {per[t][0] / n:.1f} lanes / {per[t][1] * 64 / n:.1f} slots" for t in TAGS if per[t][1]))
            print(f"{name} cap_mid={mid}: oracle (batched walk) {want} perms/proof; GPU cap={env}: {lanes / n:.1f} lane-perms/proof "
                  f"({lanes / n / want:.3f}x), {waves * 64 / n:.1f} wave-slot perms/proof ({waves * 64 / n / want:.3f}x)")


def mixed():
    """the bench workloads: round-robin mixes, where one wavefront used to see several tree geometries"""
    mixes = {"standard": ["recursive_proof_16_15.bin", "level3-1.bin", "level6-1.bin", "level7-1.bin"],
             "chain": ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin", "level7-1.bin",
                       "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin", "level13-1.bin"]}
    rsv.set_default_option("tree_cap", "on")
    for label, names in mixes.items():
        proofs = [open(os.path.join(ROOT, "tests", "golden", "proofs", f), "rb").read() for f in names]
        want = sum(ob.perm_count(p) for p in proofs) / len(proofs)
        n = 4160 if label.startswith("chain") else 4096
        counter()
        acc, _ = rsv.verify_batch([proofs[i % len(proofs)] for i in range(n)], [ob.header_cfg(proofs[i % len(proofs)]) for i in range(n)])
        assert acc.all()
        per = counter()
        lanes, waves = sum(a for a, _ in per), sum(b for _, b in per)
        print("   " + ", ".join(f"{TAGS[t]} {per[t][0] / n:.1f} lanes / {per[t][1] * 64 / n:.1f} slots" for t in TAGS if per[t][1]))
        print(f"{label} mix: oracle {want:.0f} perms/proof; GPU {lanes / n:.1f} lane-perms ({lanes / n / want:.3f}x), "
              f"{waves * 64 / n:.1f} wave-slot perms ({waves * 64 / n / want:.3f}x)")


def chain_shapes():
    """every shape of the recursion chain, one configuration per call (paced lane-form trees, as a large batch runs them)"""
    names = ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level5-1.bin", "level7-1.bin", "level8-1.bin", "level9-1.bin", "level10-1.bin", "level12-1.bin", "level13-1.bin"]
    rsv.set_default_option("tree_cap", "on")
    rsv.set_default_option("transcript_form", "lane")
    rsv.set_default_option("tree_pace", "paced")
    for name in names:
        proof = open(os.path.join(ROOT, "tests", "golden", "proofs", name), "rb").read()
        want = ob.perm_count(proof)
        n = 3840
        for mid in ("off", "auto"):
            rsv.set_default_option("cap_mid", mid)
            counter()
            acc, _ = rsv.verify_batch([proof] * n, ob.header_cfg(proof))
            assert acc.all()
            per = counter()
            lanes, waves = sum(a for a, _ in per), sum(b for _, b in per)
            print(f"{name} cap_mid={mid}: oracle {want}; lanes {lanes / n:.1f} ({lanes / n / want:.3f}x), slots {waves * 64 / n:.1f} ({waves * 64 / n / want:.3f}x)  "
                  + ", ".join(f"{TAGS[t][2:]} {per[t][0] / n:.0f}/{per[t][1] * 64 / n:.0f}" for t in TAGS if per[t][1]))
    rsv.set_default_option("tree_pace", 0)
    rsv.set_default_option("cap_mid", 0)


if __name__ == "__main__":
    mixed()
    if len(sys.argv) > 1 and sys.argv[1] == "chain":
        chain_shapes()
    else:
        main()
