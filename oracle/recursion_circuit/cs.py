"""TEST INFRASTRUCTURE (oracle): the reference's Plonk-with-Poseidon constraint system and its field variables, restated.

What the reference's recursion circuit leaves behind for the next prover is `variables: Vec<QM31>` plus the gate lists
(`a_wire / b_wire / c_wire / op / poseidon_wire / enforce_c_m31`) and the PoseidonFlow
(constraint_system/src/plonk_with_poseidon.rs:17-41).  Every gadget call appends to them in program order, so the
vector is only reproduced by replaying the gadgets in that order; this module does that with plain Python integers and
keeps for every variable HOW it came to be (`origin`), from which program.py derives a witness program.  It is an
independent restatement: the library's own mirror of the gadgets is C++ (recursive-stwo_amd/csrc/circuit_*.hpp).
Only tests/ and the parity tooling import this package; the product never does.

Values: M31 = int, CM31 = (re, im), QM31 = (a0, a1, a2, a3) = (a0 + a1 i) + (a2 + a3 i) j, i^2 = -1, j^2 = 2 + i.
"""
from __future__ import annotations

P = (1 << 31) - 1

ZERO4 = (0, 0, 0, 0)
ONE4 = (1, 0, 0, 0)
I4 = (0, 1, 0, 0)
J4 = (0, 0, 1, 0)


def m_inv(a: int) -> int:
    return pow(a, P - 2, P)


def c_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def c_inv(a):
    d = m_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (P - a[1]) * d % P)


def q_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P, (a[2] + b[2]) % P, (a[3] + b[3]) % P)


def q_neg(a):
    return ((P - a[0]) % P, (P - a[1]) % P, (P - a[2]) % P, (P - a[3]) % P)


def q_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P, (a[2] - b[2]) % P, (a[3] - b[3]) % P)


def q_mul(a, b):
    # (A + B j)(C + D j) = AC + (2 + i) BD + (AD + BC) j
    a0, a1, a2, a3 = a
    b0, b1, b2, b3 = b
    ac0, ac1 = a0 * b0 - a1 * b1, a0 * b1 + a1 * b0
    bd0, bd1 = a2 * b2 - a3 * b3, a2 * b3 + a3 * b2
    ad0, ad1 = a0 * b2 - a1 * b3, a0 * b3 + a1 * b2
    bc0, bc1 = a2 * b0 - a3 * b1, a2 * b1 + a3 * b0
    return ((ac0 + 2 * bd0 - bd1) % P, (ac1 + 2 * bd1 + bd0) % P, (ad0 + bc0) % P, (ad1 + bc1) % P)


def q_scale(a, k: int):
    return (a[0] * k % P, a[1] * k % P, a[2] * k % P, a[3] * k % P)


def q_inv(a):
    # 1 / (A + B j) = (A - B j) / (A^2 - (2 + i) B^2)
    A, B = (a[0], a[1]), (a[2], a[3])
    b2 = c_mul(B, B)
    ib2 = ((2 * b2[0] - b2[1]) % P, (2 * b2[1] + b2[0]) % P)
    a2 = c_mul(A, A)
    den = c_inv(((a2[0] - ib2[0]) % P, (a2[1] - ib2[1]) % P))
    r0 = c_mul(A, den)
    r1 = c_mul(((P - B[0]) % P, (P - B[1]) % P), den)
    return (r0[0], r0[1], r1[0], r1[1])


def q_pow(a, e: int):
    r = ONE4
    while e:
        if e & 1:
            r = q_mul(r, a)
        a = q_mul(a, a)
        e >>= 1
    return r


def q_of(x) -> tuple:
    """Any of the three value kinds as a QM31."""
    if isinstance(x, int):
        return (x % P, 0, 0, 0)
    if len(x) == 2:
        return (x[0], x[1], 0, 0)
    return x


WITNESS, CONSTANT, PUBLIC = "witness", "constant", "public"


class ConstraintSystem:
    """PlonkWithPoseidonConstraintSystem (constraint_system/src/plonk_with_poseidon.rs:17-283).  `origin[k]` says how
    variable k came to be — ("const",), ("add", a, b), ("mul", a, b), ("mulc", a, c), ("hint", tag...) — which is what a
    gate-list evaluator needs beside the wires (the reference computes the value in place instead)."""

    def __init__(self):
        self.variables = [ZERO4, ONE4, I4, J4]
        self.origin = [("const",)] * 4
        self.cache = {}
        self.a_wire, self.b_wire, self.c_wire = [0, 1, 2, 3], [0, 0, 0, 0], [0, 1, 2, 3]
        self.poseidon_wire, self.enforce_c_m31, self.op = [0] * 4, [0] * 4, [1] * 4
        self.flow = []  # (wire1, hash1, wire2, hash2, wire3, hash3, wire4, hash4, swap_addr, swap)
        self.num_input = 3
        self.hint_tag = None  # the gadget allocating a witness says what it is (program.py turns the tags into instructions)

    # -- rows
    def _row(self, a, b, c, op, pw=0, m31=0):
        self.a_wire.append(a); self.b_wire.append(b); self.c_wire.append(c)
        self.poseidon_wire.append(pw); self.enforce_c_m31.append(m31); self.op.append(op)

    def insert_gate(self, a, b, c, op):
        n = len(self.variables)
        assert a < n and b < n and c < n
        self._row(a, b, c, op)

    def enforce_zero(self, var):
        self._row(var, 0, 0, 1)

    def _push(self, value, origin):
        self.variables.append(value)
        self.origin.append(origin)
        return len(self.variables) - 1

    def add(self, a, b):
        c = self._push(q_add(self.variables[a], self.variables[b]), ("add", a, b))
        self.insert_gate(a, b, c, 1)
        return c

    def mul(self, a, b):
        c = self._push(q_mul(self.variables[a], self.variables[b]), ("mul", a, b))
        self.insert_gate(a, b, c, 0)
        return c

    def assemble_poseidon_gate(self, a, b):
        c = self._push(q_mul(self.variables[a], self.variables[b]), ("mul", a, b))
        self._row(a, b, c, 0, pw=c)
        return c

    def mul_constant(self, a, k, program_k=None):
        """program_k: the constant the witness program uses where the reference's gate constant follows the witness
        (gadgets.pm_select) — same value for every proof, same product for the proof at hand."""
        k %= P
        c = self._push(q_scale(self.variables[a], k), ("mulc", a, k if program_k is None else program_k % P))
        self.insert_gate(a, 0, c, k)
        return c

    def _origin_of_new(self, mode, value):
        if mode == CONSTANT:
            return ("const",)
        tag, self.hint_tag = self.hint_tag, None
        return ("hint", tag) if tag is not None else ("hint", None)

    def new_m31(self, v, mode):
        v %= P
        c = self._push((v, 0, 0, 0), self._origin_of_new(mode, v))
        if mode == PUBLIC:
            self._row(c, 0, c, 1, m31=1)
            self.num_input += 1
        elif mode == WITNESS:
            self._row(c, 0, c, 1, m31=1)
        else:
            self._row(1, 0, c, v)
        return c

    def new_qm31(self, v, mode):
        c = self._push(v, self._origin_of_new(mode, v))
        if mode == PUBLIC:
            self._row(c, 0, c, 1, m31=1)
            self.num_input += 1
        elif mode == CONSTANT:
            a0 = self.new_m31(v[0], CONSTANT)
            a1 = self.new_m31(v[1], CONSTANT)
            a2 = self.new_m31(v[2], CONSTANT)
            a3 = self.new_m31(v[3], CONSTANT)
            t = self.mul(a1, 2)
            a = self.add(a0, t)
            t = self.mul(a3, 2)
            t = self.add(a2, t)
            b = self.mul(t, 3)
            self._row(a, b, c, 1)
        return c

    def invoke_poseidon_accelerator(self, e1, e2, e3, e4, swap_addr, swap):
        self.flow.append((e1, e2, e3, e4, swap_addr, swap))

    # -- closing steps (plonk_with_poseidon.rs:283-466)
    def num_plonk_rows(self):
        return len(self.a_wire)

    def check_arithmetics(self):
        v = self.variables
        for i in range(len(self.a_wire)):
            a, b, c, op = v[self.a_wire[i]], v[self.b_wire[i]], v[self.c_wire[i]], self.op[i]
            want = q_add(q_scale(q_add(a, b), op), q_scale(q_mul(a, b), (1 - op) % P))
            assert c == want, (i, self.a_wire[i], self.b_wire[i], self.c_wire[i], op)
            if self.enforce_c_m31[i]:
                assert c[1] == 0 and c[2] == 0 and c[3] == 0, i

    def check_poseidon_invocations(self, permute):
        halves = {}
        for i in range(len(self.a_wire)):
            if self.poseidon_wire[i]:
                l, r = self.variables[self.a_wire[i]], self.variables[self.b_wire[i]]
                halves[self.poseidon_wire[i]] = tuple(l) + tuple(r)
        for (e1, e2, e3, e4, _addr, swap) in self.flow:
            for w, h in (e1, e2, e3, e4):
                if w:
                    assert halves[w] == tuple(h), w
            state = list(e2[1]) + list(e1[1]) if swap else list(e1[1]) + list(e2[1])
            assert tuple(permute(state)) == tuple(e3[1]) + tuple(e4[1])


class Var:
    """M31Var / CM31Var / QM31Var (primitives/fields/src/{m31,cm31,qm31}.rs): a value and the index of its variable.
    `kind` is 1, 2 or 4 (how many M31 coordinates the value has); arithmetic between kinds follows the reference's
    operator impls, which all come down to one cs.add / cs.mul / cs.mul_constant on the QM31 variables."""
    __slots__ = ("cs", "value", "variable", "kind")

    def __init__(self, cs, value, variable, kind):
        self.cs, self.value, self.variable, self.kind = cs, value, variable, kind

    def q(self):
        return q_of(self.value)


def _narrow(q, kind):
    if kind == 1:
        return q[0]
    if kind == 2:
        return (q[0], q[1])
    return q


def add(a: Var, b: Var) -> Var:
    kind = max(a.kind, b.kind)
    if a.kind < b.kind:  # `&M31Var + &QM31Var` and its likes are written `rhs + self` in the reference: the wires swap
        a, b = b, a
    return Var(a.cs, _narrow(q_add(a.q(), b.q()), kind), a.cs.add(a.variable, b.variable), kind)


def neg(a: Var) -> Var:
    return Var(a.cs, _narrow(q_neg(a.q()), a.kind), a.cs.mul_constant(a.variable, P - 1), a.kind)


def sub(a: Var, b: Var) -> Var:
    """self + &(-rhs)."""
    return add(a, neg(b))


def mul(a: Var, b: Var) -> Var:
    kind = max(a.kind, b.kind)
    if a.kind < b.kind:  # `rhs * self`
        a, b = b, a
    return Var(a.cs, _narrow(q_mul(a.q(), b.q()), kind), a.cs.mul(a.variable, b.variable), kind)


def mul_constant(a: Var, k: int) -> Var:
    return Var(a.cs, _narrow(q_scale(a.q(), k % P), a.kind), a.cs.mul_constant(a.variable, k % P), a.kind)


def equalverify(a: Var, b: Var):
    assert a.q() == b.q(), (a.value, b.value)
    a.cs.insert_gate(a.variable, 0, b.variable, 1)


# -- M31Var (m31.rs)
def m31_zero(cs): return Var(cs, 0, 0, 1)
def m31_one(cs): return Var(cs, 1, 1, 1)


def m31_constant(cs, v):
    v %= P
    if v == 0:
        return m31_zero(cs)
    if v == 1:
        return m31_one(cs)
    key = "m31 %d" % v
    if key not in cs.cache:
        cs.cache[key] = cs.new_m31(v, CONSTANT)
    return Var(cs, v, cs.cache[key], 1)


def m31_witness(cs, v, tag=None):
    cs.hint_tag = tag
    return Var(cs, v % P, cs.new_m31(v, WITNESS), 1)


def m31_inv(a: Var) -> Var:
    res = m31_witness(a.cs, m_inv(a.value), ("inv", a.variable))
    a.cs.insert_gate(a.variable, res.variable, 1, 0)
    return res


def m31_is_zero(a: Var) -> Var:
    cs = a.cs
    inv = m31_witness(cs, 0 if a.value == 0 else m_inv(a.value), ("inv_or_zero", a.variable))
    out = add(neg(mul(a, inv)), m31_one(cs))
    cs.insert_gate(a.variable, out.variable, 0, 0)
    return out


def m31_is_eq(a: Var, b: Var) -> Var:
    return m31_is_zero(sub(a, b))


# -- CM31Var (cm31.rs)
def cm31_from_m31(real: Var, imag: Var) -> Var:
    cs = real.cs
    return Var(cs, (real.value, imag.value), cs.add(real.variable, cs.mul(imag.variable, 2)), 2)


def cm31_witness(cs, v, tag=None):
    real = m31_witness(cs, v[0], tag and tag + (0,))
    imag = m31_witness(cs, v[1], tag and tag + (1,))
    return cm31_from_m31(real, imag)


def cm31_constant(cs, v):
    v = (v[0] % P, v[1] % P)
    if v == (0, 0):
        return Var(cs, v, 0, 2)
    if v == (1, 0):
        return Var(cs, v, 1, 2)
    if v == (0, 1):
        return Var(cs, v, 2, 2)
    key = "cm31 %d,%d" % v
    if key not in cs.cache:
        real, imag = m31_constant(cs, v[0]), m31_constant(cs, v[1])
        cs.cache[key] = cs.add(real.variable, cs.mul(imag.variable, 2))
    return Var(cs, v, cs.cache[key], 2)


def cm31_inv(a: Var) -> Var:
    return cm31_witness(a.cs, c_inv(a.value), ("cinv", a.variable))  # no gate ties it to `a` (cm31.rs:240-246)


def shift_by_i(a: Var) -> Var:
    kind = max(a.kind, 2)
    return Var(a.cs, _narrow(q_mul(a.q(), I4), kind), a.cs.mul(a.variable, 2), kind)


def shift_by_j(a: Var) -> Var:
    return Var(a.cs, q_mul(a.q(), J4), a.cs.mul(a.variable, 3), 4)


def shift_by_ij(a: Var) -> Var:
    return shift_by_j(shift_by_i(a))


def mul_constant_cm31(a: Var, k) -> Var:
    """CM31Var / QM31Var::mul_constant_cm31."""
    cs = a.cs
    x = mul_constant(a, k[0])
    y = mul_constant(a, k[1])
    kind = max(a.kind, 2)
    return Var(cs, _narrow(q_mul(a.q(), (k[0] % P, k[1] % P, 0, 0)), kind), cs.add(x.variable, cs.mul(y.variable, 2)), kind)


# -- QM31Var (qm31.rs)
def qm31_zero(cs): return Var(cs, ZERO4, 0, 4)
def qm31_one(cs): return Var(cs, ONE4, 1, 4)
def qm31_i(cs): return Var(cs, I4, 2, 4)
def qm31_j(cs): return Var(cs, J4, 3, 4)


def qm31_witness(cs, v, tag=None):
    cs.hint_tag = tag
    return Var(cs, tuple(x % P for x in v), cs.new_qm31(tuple(x % P for x in v), WITNESS), 4)


def qm31_constant(cs, v):
    v = tuple(x % P for x in v)
    for special, idx in ((ZERO4, 0), (ONE4, 1), (I4, 2), (J4, 3)):
        if v == special:
            return Var(cs, v, idx, 4)
    key = "qm31 %d,%d,%d,%d" % v
    if key not in cs.cache:
        cs.cache[key] = cs.new_qm31(v, CONSTANT)
    return Var(cs, v, cs.cache[key], 4)


def qm31_from_m31(a0, a1, a2, a3) -> Var:
    cs = a0.cs
    l = cs.add(a0.variable, cs.mul(a1.variable, 2))
    r = cs.mul(cs.add(a2.variable, cs.mul(a3.variable, 2)), 3)
    return Var(cs, (a0.value, a1.value, a2.value, a3.value), cs.add(l, r), 4)


def qm31_from_cm31(a: Var, b: Var) -> Var:
    cs = a.cs
    return Var(cs, (a.value[0], a.value[1], b.value[0], b.value[1]), cs.add(a.variable, cs.mul(b.variable, 3)), 4)


def as_qm31(a: Var) -> Var:
    """QM31Var::from(&M31Var): same variable."""
    return Var(a.cs, a.q(), a.variable, 4)


def as_cm31(a: Var) -> Var:
    q = a.q()
    return Var(a.cs, (q[0], q[1]), a.variable, 2)


def decompose_m31(a: Var):
    cs = a.cs
    parts = [m31_witness(cs, a.value[k], ("coord", a.variable, k)) for k in range(4)]
    l = cs.add(parts[0].variable, cs.mul(parts[1].variable, 2))
    r = cs.mul(cs.add(parts[2].variable, cs.mul(parts[3].variable, 2)), 3)
    cs.insert_gate(l, r, a.variable, 1)
    return parts


def decompose_cm31(a: Var):
    v = decompose_m31(a)
    a0 = add(shift_by_i(as_cm31(v[1])), v[0])
    a1 = add(shift_by_i(as_cm31(v[3])), v[2])
    return [a0, a1]


def qm31_inv(a: Var) -> Var:
    res = qm31_witness(a.cs, q_inv(a.value), ("qinv", a.variable))
    a.cs.insert_gate(a.variable, res.variable, 1, 0)
    return res


def qm31_pow(a: Var, exp: int) -> Var:
    bools = []
    while exp > 0:
        bools.append(exp & 1)
        exp >>= 1
    cur = qm31_one(a.cs)
    for i in range(len(bools) - 1, -1, -1):
        if bools[i]:
            cur = mul(cur, a)
        if i != 0:
            cur = mul(cur, cur)
    return cur


def mul_constant_qm31(a: Var, k) -> Var:
    cs = a.cs
    kv = cs.new_qm31(tuple(x % P for x in k), CONSTANT)  # not cached (qm31.rs:389-398)
    return Var(cs, q_mul(a.q(), tuple(x % P for x in k)), cs.mul(a.variable, kv), 4)


def select(a: Var, b: Var, bit_value, bit_variable) -> Var:
    cs = a.cs
    d = sub(b, a)
    v = cs.mul(d.variable, bit_variable)
    v = cs.add(a.variable, v)
    return Var(cs, b.value if bit_value else a.value, v, a.kind)


def swap(a: Var, b: Var, bit_value, bit_variable):
    cs = a.cs
    d = sub(b, a)
    left = cs.mul(d.variable, bit_variable)
    right = cs.mul_constant(left, P - 1)
    left = cs.add(a.variable, left)
    right = cs.add(b.variable, right)
    lv, rv = (b.value, a.value) if bit_value else (a.value, b.value)
    return Var(cs, lv, left, a.kind), Var(cs, rv, right, b.kind)
