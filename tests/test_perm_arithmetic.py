"""CPU checks of arithmetic identities the HIP permutation relies on (recursive-stwo_amd/csrc/poseidon2.hpp), over the
round constants of the parameter set (read from the oracle's table, primitives/poseidon31/src/parameters.rs:6-190)."""
import ctypes

import numpy as np

from tests import oracle_binding as ob

P = 0x7FFFFFFF
M32 = 0xFFFFFFFF


def _constants():
    ob.lib.rsvo_round_constants.restype = ctypes.POINTER(ctypes.c_uint32)
    ob.lib.rsvo_round_constants.argtypes = [ctypes.c_int]
    first = [ob.lib.rsvo_round_constants(0)[i] for i in range(64)]
    partial = [ob.lib.rsvo_round_constants(1)[i] for i in range(14)]
    last = [ob.lib.rsvo_round_constants(2)[i] for i in range(64)]
    return first + partial + last


def test_fused_round_constant_reduction():
    """canon_rc: for a lazily folded t <= P + HI (HI < 2^19) and c = P - rc, min(t - c, t - c + P) in 32-bit wrapping
    arithmetic is the canonical (t + rc) mod P — provided every rc < P - 2^19, which the kernel static_asserts."""
    rcs = _constants()
    assert len(rcs) == 142 and max(rcs) < P - (1 << 19)
    hi = (1 << 19) - 1
    rng = np.random.default_rng(3)
    for rc in rcs:
        c = P - rc
        ts = [0, 1, c - 1, c, c + 1, P - 1, P, P + 1, P + 160, P + hi] + [int(x) for x in rng.integers(0, P + hi + 1, 64)]
        for t in ts:
            a = (t - c) & M32
            b = (a + P) & M32
            got = min(a, b)
            assert got == (t + rc) % P and got < P, (hex(rc), t)


def test_doubled_fold():
    """fold2: for V = 2v (v < 2^62), hi32(V) + (lo32(V) >> 1) == (v >> 31) + (v & P), a value congruent to v mod P that
    fits 32 bits."""
    rng = np.random.default_rng(4)
    vs = [0, 1, P, P + 1, (1 << 62) - 1] + [int(x) for x in rng.integers(0, 1 << 62, 2000, dtype=np.uint64)]
    for v in vs:
        V = 2 * v
        got = (V >> 32) + ((V & M32) >> 1)
        assert got == (v >> 31) + (v & P) and got % P == v % P and got <= M32
