"""TEST INFRASTRUCTURE (oracle): from the constraint system to the Plonk component's trace columns, and the value of a
column's interpolant at an out-of-domain point — the link between a restated circuit and the NEXT proof of the
reference's fixture chain, which carries those values as `sampled_values` (tests/test_recursion_circuit.py).

pad / populate_logup_arguments / generate_plonk_with_poseidon_circuit: constraint_system/src/plonk_with_poseidon.rs:283-
466, 522-629.  Column order inside the trees: components/recursive/composition/src/plonk.rs:14-41 (the order the AIR
reads them in).  Domain conventions (stwo core, not vendored; each is one line of published stwo): the trace of log size
n lives on CanonicCoset(n).circle_domain() = half_odds(n - 1) and its conjugate, stored in bit-reversed order."""
from __future__ import annotations

import numpy as np


from . import cs as C
from . import gadgets as G
from .cs import P

PREPROCESSED = ("a_wire", "b_wire", "c_wire", "op", "mult_a", "mult_b", "mult_c", "poseidon_wire", "mult_poseidon", "enforce_c_m31")


def pad(cs):
    """-> number of Plonk rows after padding.  The flow is padded first, to a multiple of 16 and at least 32 entries, with
    invocations of constant hashes whose wires and swap address are variable 0."""
    n_flow = len(cs.flow)
    for _ in range(n_flow, max(32, -(-n_flow // 16) * 16)):
        cs.flow.append(((0, None), (0, None), (0, None), (0, None), 0, False))
    n = len(cs.a_wire)
    padded = 1 << (n - 1).bit_length()
    for _ in range(n, padded):
        cs._row(0, 0, 0, 1)
    return padded


def populate_logup_arguments(cs):
    n_vars, n_rows = len(cs.variables), len(cs.a_wire)
    counts = np.zeros(n_vars, np.int64)
    a, b, c = (np.array(x, dtype=np.int64) for x in (cs.a_wire, cs.b_wire, cs.c_wire))
    for w in (a, b, c):
        np.add.at(counts, w, 1)
    counts[1:1 + cs.num_input] += 1
    for (_e1, _e2, _e3, _e4, addr, _sw) in cs.flow:
        counts[addr] += 1
    first = np.zeros(n_vars, bool)
    mult = {k: np.ones(n_rows, np.int64) for k in "abc"}
    for i in range(n_rows):  # first occurrence in row order a, b, c
        for k, w in (("a", cs.a_wire[i]), ("b", cs.b_wire[i]), ("c", cs.c_wire[i])):
            if not first[w]:
                first[w] = True
                mult[k][i] = 1 - counts[w]
    mp_vars = np.zeros(n_vars, np.int64)
    for (e1, e2, e3, e4, _addr, _sw) in cs.flow:
        for w, _h in (e1, e2, e3, e4):
            mp_vars[w] += 1
    mp_vars[0] = 0
    mult_poseidon = np.zeros(n_rows, np.int64)
    for i in range(n_rows):
        w = cs.poseidon_wire[i]
        if mp_vars[w]:
            mult_poseidon[i] = mp_vars[w]
            assert counts[w] == 1
            mp_vars[w] = 0
    return mult["a"] % P, mult["b"] % P, mult["c"] % P, mult_poseidon


def plonk_columns(cs):
    """-> (preprocessed: dict name -> int64[n], trace: int64[12, n]) after pad()."""
    mult_a, mult_b, mult_c, mult_poseidon = populate_logup_arguments(cs)
    arr = lambda x: np.array(x, dtype=np.int64) % P
    pre = {"a_wire": arr(cs.a_wire), "b_wire": arr(cs.b_wire), "c_wire": arr(cs.c_wire), "op": arr(cs.op), "mult_a": mult_a,
           "mult_b": mult_b, "mult_c": mult_c, "poseidon_wire": arr(cs.poseidon_wire), "mult_poseidon": mult_poseidon,
           "enforce_c_m31": arr(cs.enforce_c_m31)}
    v = np.array(cs.variables, dtype=np.int64)
    trace = np.concatenate([v[np.array(w)].T for w in (cs.a_wire, cs.b_wire, cs.c_wire)])
    return pre, trace


# ---------------------------------------------------------------- interpolant of a column at one point
def _bit_reverse_perm(log_n):
    idx = np.arange(1 << log_n, dtype=np.int64)
    rev = np.zeros_like(idx)
    for b in range(log_n):
        rev |= ((idx >> b) & 1) << (log_n - 1 - b)
    return rev


def _coset_points(initial_index, step_index, log_size):
    """x, y (int64 arrays, natural order) of initial + k * step, k < 2^log_size."""
    x = np.array([G.point_of_index(initial_index)[0]], dtype=object)
    y = np.array([G.point_of_index(initial_index)[1]], dtype=object)
    for j in range(log_size):
        sx, sy = G.point_of_index(step_index << j)
        nx = (x * sx - y * sy) % P
        ny = (x * sy + y * sx) % P
        x, y = np.concatenate([x, nx]), np.concatenate([y, ny])
    return x, y


class PointEvaluator:
    """f(point) for columns given on CanonicCoset(log_size).circle_domain() in bit-reversed order: the circle-to-line
    and line folds of the interpolant with the point's coordinates in the place of the folding randomness.  The weights
    of all rows are computed once (f(point) = sum_i w_i v_i), so a column costs one dot product."""

    def __init__(self, log_size, point):
        n = log_size
        half = G.canonic_half_coset(n)
        hx, hy = _coset_points(half.initial_index, half.step_size, n - 1)
        rev = _bit_reverse_perm(n - 1)
        xs, ys = hx[rev], hy[rev]
        px, py = tuple(point[0]), tuple(point[1])
        inv2 = pow(2, P - 2, P)
        # run the folds on the identity: weights[i] = coefficient of v_i in f(point)
        # backwards: start from the last fold's two weights and expand
        stages = []  # per line stage: (inv_2x array over pairs, X)
        X = px
        cur_x = xs
        while len(cur_x) > 1:
            ev = cur_x[0::2]
            stages.append((np.array([pow(int(t), P - 2, P) * inv2 % P for t in ev], dtype=object), X))
            cur_x = (2 * ev * ev - 1) % P
            X = C.q_sub(C.q_scale(C.q_mul(X, X), 2), (1, 0, 0, 0))
        w = [np.array([1], dtype=object), np.array([0], dtype=object), np.array([0], dtype=object), np.array([0], dtype=object)]
        for inv2x, X in reversed(stages):
            # g' = (a + b)/2 + X (a - b) inv2x  ->  w_a = w' (1/2 + X inv2x), w_b = w' (1/2 - X inv2x)
            t = [np.broadcast_to(np.array(X[k], dtype=object), inv2x.shape) * inv2x % P for k in range(4)]
            plus = [(t[0] + inv2) % P, t[1], t[2], t[3]]
            minus = [(inv2 - t[0]) % P, (-t[1]) % P, (-t[2]) % P, (-t[3]) % P]
            wa = self._q_mul_arr(w, plus)
            wb = self._q_mul_arr(w, minus)
            w = [np.stack([wa[k], wb[k]], axis=1).reshape(-1) for k in range(4)]
        # circle-to-line stage: g = (a + b)/2 + py (a - b) / (2 y)
        inv2y = np.array([pow(int(t), P - 2, P) * inv2 % P for t in ys], dtype=object)
        t = [np.broadcast_to(np.array(py[k], dtype=object), inv2y.shape) * inv2y % P for k in range(4)]
        plus = [(t[0] + inv2) % P, t[1], t[2], t[3]]
        minus = [(inv2 - t[0]) % P, (-t[1]) % P, (-t[2]) % P, (-t[3]) % P]
        wa = self._q_mul_arr(w, plus)
        wb = self._q_mul_arr(w, minus)
        self.weights = [np.stack([wa[k], wb[k]], axis=1).reshape(-1) for k in range(4)]
        self.log_size = n

    @staticmethod
    def _q_mul_arr(a, b):
        a0, a1, a2, a3 = a
        b0, b1, b2, b3 = b
        ac0, ac1 = a0 * b0 - a1 * b1, a0 * b1 + a1 * b0
        bd0, bd1 = a2 * b2 - a3 * b3, a2 * b3 + a3 * b2
        ad0, ad1 = a0 * b2 - a1 * b3, a0 * b3 + a1 * b2
        bc0, bc1 = a2 * b0 - a3 * b1, a2 * b1 + a3 * b0
        return [(ac0 + 2 * bd0 - bd1) % P, (ac1 + 2 * bd1 + bd0) % P, (ad0 + bc0) % P, (ad1 + bc1) % P]

    def eval(self, column):
        """column: M31 values at the bit-reversed positions, length 2^log_size."""
        col = np.array(column, dtype=object)
        return tuple(int((self.weights[k] * col).sum() % P) for k in range(4))


# ---------------------------------------------------------------- the Poseidon component's columns
# The component's trace generator lives in the un-vendored stwo fork; what follows is what the AIR the verifier
# evaluates (components/recursive/composition/src/poseidon.rs:73-241) forces, row by row: six rows per invocation —
# [0] is_first: in = the two input halves as given, intermediate[0] = the swap bit, out = external matrix of the
# (swapped) input; [1], [2] is_full: two full rounds each, intermediate = the first S-box layer; [3] partial: the 14
# partial rounds, intermediate[r] = the r-th S-box output; [4], [5] is_full, [5] is_last.  Rows hand their state on
# through the lookup ids 2 * round_id (+ 0..3), so round_id counts the rounds of the whole flow (6 k + j).
# What the AIR does not say was found against the fixtures (tests/test_recursion_circuit.py): sixteen invocations share
# a block of 96 rows, round-major (row = ((k / 16) * 6 + j) * 16 + k % 16 — the SIMD backend's 16 lanes), the words the
# AIR leaves free are zero, and the rows behind the padded flow have is_first_round = is_last_round = 1.
def _mds4(x):
    t0, t1 = x[0] + x[1], x[2] + x[3]
    t2, t3 = 2 * x[1] + t1, 2 * x[3] + t0
    t4, t5 = 4 * t1 + t3, 4 * t0 + t2
    return [(t3 + t5) % P, t5 % P, (t2 + t4) % P, t4 % P]


def _ext(s):
    s = [v for g in range(4) for v in _mds4(s[4 * g:4 * g + 4])]
    sums = [(s[j] + s[j + 4] + s[j + 8] + s[j + 12]) % P for j in range(4)]
    return [(s[i] + sums[i % 4]) % P for i in range(16)]


def _pow5(x):
    x2 = x * x % P
    return x2 * x2 % P * x % P


def poseidon_columns(flow, round_constants, log_size, padding_hash=None):
    """flow: cs.flow after pad(); round_constants: (first[4][16], partial[14], last[4][16]).
    -> (preprocessed int64[40, n], trace int64[48, n]), n = 2^log_size, rows beyond 6 * len(flow) zero."""
    first, partial, last = round_constants
    n = 1 << log_size
    pre = np.zeros((40, n), dtype=object)
    tr = np.zeros((48, n), dtype=object)
    for k, (e1, e2, e3, e4, addr, swap) in enumerate(flow):
        h1 = list(e1[1]) if e1[1] is not None else list(padding_hash[0])
        h2 = list(e2[1]) if e2[1] is not None else list(padding_hash[0])
        rid = 6 * k
        row = [((k // 16) * 6 + j) * 16 + k % 16 for j in range(6)]
        rows_in, rows_mid, rows_out = [], [], []
        state = h1 + h2
        sw = (h2 + h1) if swap else (h1 + h2)
        out = _ext(sw)
        rows_in.append(state); rows_mid.append([1 if swap else 0] + [0] * 15); rows_out.append(out)
        cur = out
        for pair in (first[0:2], first[2:4]):
            mid = [_pow5((cur[i] + pair[0][i]) % P) for i in range(16)]
            o = _ext([_pow5((v + pair[1][i]) % P) for i, v in enumerate(_ext(mid))])
            rows_in.append(cur); rows_mid.append(mid); rows_out.append(o)
            cur = o
        s = list(cur)
        mids = []
        for rr in range(14):
            s[0] = _pow5((s[0] + partial[rr]) % P)
            mids.append(s[0])
            total = sum(s) % P
            s = [(total + 3 * s[0]) % P] + [(total + (s[i] << (i + 1))) % P for i in range(1, 16)]
        rows_in.append(cur); rows_mid.append(mids + [0, 0]); rows_out.append(s)
        cur = s
        for pair in (last[0:2], last[2:4]):
            mid = [_pow5((cur[i] + pair[0][i]) % P) for i in range(16)]
            o = _ext([_pow5((v + pair[1][i]) % P) for i, v in enumerate(_ext(mid))])
            rows_in.append(cur); rows_mid.append(mid); rows_out.append(o)
            cur = o
        if e3[1] is not None:
            assert cur == list(e3[1]) + list(e4[1])
        for j in range(6):
            tr[0:16, row[j]] = rows_in[j]
            tr[16:32, row[j]] = rows_mid[j]
            tr[32:48, row[j]] = rows_out[j]
            pre[3, row[j]] = rid + j
        pre[0, row[0]] = 1
        pre[1, row[5]] = 1
        for j in (1, 2, 4, 5):
            pre[2, row[j]] = 1
        pre[4, row[0]] = addr  # rc0[0] of the first row carries the swap bit's address
        for j, pair in ((1, first[0:2]), (2, first[2:4]), (4, last[0:2]), (5, last[2:4])):
            pre[4:20, row[j]] = pair[0]
            pre[20:36, row[j]] = pair[1]
        pre[4:18, row[3]] = partial
        pre[36, row[0]], pre[37, row[0]] = e1[0], e2[0]
        pre[38, row[0]], pre[39, row[0]] = int(e1[0] != 0), int(e2[0] != 0)
        pre[36, row[5]], pre[37, row[5]] = e3[0], e4[0]
        pre[38, row[5]], pre[39, row[5]] = int(e3[0] != 0), int(e4[0] != 0)
    # rows behind the last invocation: a round that is first and last at once over an all-zero state (every constraint
    # holds, no lookup is made)
    pre[0, 6 * len(flow):] = 1
    pre[1, 6 * len(flow):] = 1
    return pre, tr
