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


# ---------------------------------------------------------------------------------------------------------------------
# The ROW form (poseidon2_row.hpp, round 4): one state per 16-lane row, values only WEAKLY reduced between steps so that
# the chain of dependent instructions is short.  The model below is the kernel's arithmetic statement by statement in
# Python integers, with every register width asserted (u32 / u64 never wrap) and the ranges the comments claim checked;
# its output must be the oracle's permutation.  (The GPU test compares the kernel itself with the oracle.)
def _u32(v):
    assert 0 <= v <= M32, hex(v)
    return v


def _u64(v):
    assert 0 <= v < (1 << 64), hex(v)
    return v


def _mulw(a, b):
    """any u32 x any u32 -> congruent value <= 2 P + 3: the 64-bit product as three 31-bit limbs (2^31 = 1 mod P)"""
    t = _u64(_u32(a) * _u32(b))
    hi, lo = t >> 32, t & M32
    B = (((hi << 32) | lo) >> 31) & M32 & P       # v_alignbit_b32(hi, lo, 31) & P
    out = _u32((lo & P) + B + (hi >> 30))
    assert out <= 2 * P + 3 and out % P == (a * b) % P
    return out


def _sbox_w(u):
    x2 = _mulw(u, u)
    return _mulw(_mulw(x2, x2), u)


def _fold(v, bound_bits, add=0):
    """v < 2^bound_bits (u64) -> (v & P) + (v >> 31) + add, which must fit 32 bits"""
    assert v < (1 << bound_bits)
    return _u32((v & P) + (v >> 31) + add)


def _mds_row_w(x, rc=None):
    """external matrix on a row of 16 lanes (any u32 each) -> weak values <= P + 2^8 (+ round constant of the next S-box)"""
    acc = []
    for i in range(16):
        q, k = i & ~3, i & 3
        ca, cb, cd = (6, 1, 4) if k & 1 else (5, 7, 3)
        a = _u64(_u32(x[i]) * ca + x[q + ((k + 2) & 3)])
        a = _u64(x[q + ((k + 1) & 3)] * cb + a)
        a = _u64(x[q + ((k + 3) & 3)] * cd + a)
        assert a < (1 << 36)
        acc.append(a)
    out = []
    for i in range(16):
        w = _u64((acc[i] << 1) + acc[(i + 4) & 15] + acc[(i + 8) & 15] + acc[(i + 12) & 15])
        f = _fold(w, 39, rc[i] if rc else 0)
        assert f <= P + (1 << 8) + (rc[i] if rc else 0)
        out.append(f)
    return out


def _poseidon2_row_model(state, consts):
    first, partial, last = consts[:64], consts[64:78], consts[78:]
    full = [first[16 * r:16 * r + 16] for r in range(4)] + [last[16 * r:16 * r + 16] for r in range(4)]
    x = _mds_row_w(list(state), full[0])                       # x = M s + rc of the first S-box layer
    for r in range(4):
        x = [_sbox_w(v) for v in x]                            # <= 2 P + 3
        x = _mds_row_w(x, full[r + 1] if r < 3 else None)
    # partial rounds: u = S-box input of word 0 (every lane holds it)
    u = _u32(x[0] + partial[0])
    for r in range(14):
        s = _sbox_w(u)
        xm = [0] + x[1:]
        sl, sh = sum(v & 0xFFFF for v in xm), sum(v >> 16 for v in xm)
        assert sl < (1 << 20) and sh < (1 << 20)
        rest = _u64((sh << 16) + sl)
        assert rest < (1 << 36) and rest == sum(xm)
        v0 = _u64(s * 4 + rest)
        new = [_fold(v0, 50)]
        for i in range(1, 16):
            pre = _u64(xm[i] * (1 << (i + 1)) + rest)
            new.append(_fold(_u64(s + pre), 50))
        x = new
        assert max(x) <= P + (1 << 19)
        if r + 1 < 14:
            u = _fold(v0, 50, partial[r + 1])
    x = [_u32(v + c) for v, c in zip(x, full[4])]              # <= P + 2^19 + rc < 2^32 (every rc < P - 2^19)
    for r in range(4, 8):
        x = [_sbox_w(v) for v in x]
        x = _mds_row_w(x, full[r + 1] if r < 7 else None)
    out = []
    for t in x:                                                # canonical: min(t, t - P) in wrapping 32-bit arithmetic
        out.append(min(t, (t - P) & M32))
    return out


def test_row_form_weak_range_model():
    consts = _constants()
    rng = np.random.default_rng(11)
    states = [list(range(16)), [0] * 16, [P - 1] * 16, [P] * 16, [M32] * 16,   # incl. non-canonical garbage: still congruent
              [P - 1 if i % 2 else 0 for i in range(16)]]
    states += [[int(v) for v in rng.integers(0, P, 16)] for _ in range(40)]
    states += [[int(v) for v in rng.integers(0, 1 << 32, 16)] for _ in range(10)]
    for st in states:
        want = ob.poseidon2_permute(np.array([v % P for v in st], dtype=np.uint32)).reshape(-1).tolist()
        assert _poseidon2_row_model(st, consts) == want, st
