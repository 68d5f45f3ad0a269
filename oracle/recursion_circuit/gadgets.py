"""TEST INFRASTRUCTURE (oracle): the reference's circuit gadgets over cs.py, restated in allocation order.

bits (primitives/bits/src/lib.rs), Poseidon2HalfVar (primitives/poseidon31/src/lib.rs), ChannelVar
(primitives/channel/src/lib.rs), Poseidon31MerkleHasherVar (primitives/merkle/src/lib.rs), circle points
(primitives/circle/src/lib.rs), LinePolyVar (primitives/line/src/lib.rs), query positions (primitives/query/src/lib.rs).
"""
from __future__ import annotations

from . import cs as C
from .cs import P, Var

# ---------------------------------------------------------------- the M31 circle group (stwo core/circle.rs; published)
GEN = (2, 1268011823)  # generator of the circle group of order 2^31


def cp_add(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def cp_double(a):
    return cp_add(a, a)


def cp_neg(a):
    return (a[0], (P - a[1]) % P)


def cp_mul(a, k: int):
    k %= 1 << 31
    r = (1, 0)
    while k:
        if k & 1:
            r = cp_add(r, a)
        a = cp_double(a)
        k >>= 1
    return r


def point_of_index(idx: int):
    """CirclePointIndex(idx).to_point()."""
    return cp_mul(GEN, idx)


def subgroup_gen_index(log_size: int) -> int:
    return 1 << (31 - log_size)


class Coset:
    """stwo Coset: initial_index + i * step_size, i < 2^log_size (indices mod 2^31)."""

    def __init__(self, initial_index, log_size):
        self.initial_index = initial_index % (1 << 31)
        self.step_size = subgroup_gen_index(log_size)
        self.log_size = log_size
        self.initial = point_of_index(self.initial_index)
        self.step = point_of_index(self.step_size)

    @staticmethod
    def odds(log_size):
        return Coset(subgroup_gen_index(log_size + 1), log_size)

    @staticmethod
    def half_odds(log_size):
        return Coset(subgroup_gen_index(log_size + 2), log_size)


def canonic_coset(log_size):
    """CanonicCoset::new(log_size).coset = Coset::odds(log_size)."""
    return Coset.odds(log_size)


def canonic_half_coset(log_size):
    """CanonicCoset::new(log_size).circle_domain().half_coset = Coset::half_odds(log_size - 1)."""
    return Coset.half_odds(log_size - 1)


# ---------------------------------------------------------------- bits
class Bits:
    __slots__ = ("cs", "value", "variables")

    def __init__(self, cs, value, variables):
        self.cs, self.value, self.variables = cs, list(value), list(variables)

    def get_value(self):
        return sum(1 << k for k, b in enumerate(self.value) if b) % P

    def index_range(self, lo, hi):
        return Bits(self.cs, self.value[lo:hi], self.variables[lo:hi])

    def index_range_from(self, lo):
        return Bits(self.cs, self.value[lo:], self.variables[lo:])

    def compose_range(self, lo, hi) -> Var:
        cs = self.cs
        total = 1 if self.value[lo] else 0
        var = self.variables[lo]
        for shift, i in enumerate(range(lo + 1, hi)):
            if self.value[i]:
                total += 1 << (shift + 1)
            shifted = cs.mul_constant(self.variables[i], 1 << (shift + 1))
            var = cs.add(var, shifted)
        return Var(cs, total % P, var, 1)


def bits_witness(cs, bools, tag=None):
    variables = []
    for k, b in enumerate(bools):
        cs.hint_tag = tag and tag + (k,)
        bit = cs.new_qm31((1, 0, 0, 0) if b else C.ZERO4, C.WITNESS)
        variables.append(bit)
        minus_one = C.m31_constant(cs, P - 1)
        bit_minus_one = cs.add(bit, minus_one.variable)
        cs.insert_gate(bit, bit_minus_one, 0, 0)
    return Bits(cs, bools, variables)


def bits_from_m31(v: Var, l: int) -> Bits:
    cs = v.cs
    bools = [bool((v.value >> k) & 1) for k in range(l)]
    res = bits_witness(cs, bools, ("bit", v.variable))
    rec = Var(cs, 1 if res.value[0] else 0, res.variables[0], 1)
    for i in range(1, l):
        term = C.mul_constant(Var(cs, 1 if res.value[i] else 0, res.variables[i], 1), 1 << i)
        rec = C.add(rec, term)
    C.equalverify(rec, v)
    if l == 31:
        product = cs.mul(res.variables[0], res.variables[1])
        for i in range(2, l):
            product = cs.mul(product, res.variables[i])
        cs.enforce_zero(product)
    return res


# ---------------------------------------------------------------- Poseidon2HalfVar (native form)
PERMUTE = None  # set by the caller: 16 ints -> 16 ints, the Poseidon2 permutation of the state the circuit is about to permute


class Half:
    __slots__ = ("cs", "value", "left", "right", "sel")

    def __init__(self, cs, value, left, right, sel):
        self.cs, self.value, self.left, self.right, self.sel = cs, tuple(value), left, right, sel

    def to_qm31(self):
        return [Var(self.cs, self.value[0:4], self.left, 4), Var(self.cs, self.value[4:8], self.right, 4)]


def half_single_use(cs, value8):
    return Half(cs, value8, 0, 0, 0)


def half_from_m31(vars8):
    cs = vars8[0].cs
    left = C.qm31_from_m31(*vars8[0:4])
    right = C.qm31_from_m31(*vars8[4:8])
    sel = cs.assemble_poseidon_gate(left.variable, right.variable)
    return Half(cs, [v.value for v in vars8], left.variable, right.variable, sel)


def half_from_qm31(a: Var, b: Var):
    cs = a.cs
    sel = cs.assemble_poseidon_gate(a.variable, b.variable)
    return Half(cs, tuple(a.q()) + tuple(b.q()), a.variable, b.variable, sel)


def half_witness(cs, value8, tag=None):
    left = C.qm31_witness(cs, value8[0:4], tag and tag + (0,))
    right = C.qm31_witness(cs, value8[4:8], tag and tag + (1,))
    sel = cs.assemble_poseidon_gate(left.variable, right.variable)
    return Half(cs, value8, left.variable, right.variable, sel)


def half_zero(cs):
    if "poseidon2 zero_half" not in cs.cache:
        cs.cache["poseidon2 zero_half"] = cs.assemble_poseidon_gate(0, 0)
    return Half(cs, (0,) * 8, 0, 0, cs.cache["poseidon2 zero_half"])


def permute(left: Half, right: Half, ignore_left, ignore_right, is_swap=None):
    """Poseidon2HalfVar::permute (poseidon31/src/lib.rs:282-423).  is_swap = (bit value, bit variable) or None."""
    cs = left.cs
    swapped = is_swap is not None and is_swap[0]
    state = list(right.value) + list(left.value) if swapped else list(left.value) + list(right.value)
    out = [int(x) for x in PERMUTE(state)]
    flow_idx = len(cs.flow)

    def result(vals, ignore, half):
        if ignore:
            return Half(cs, vals, 0, 0, 0)
        l = C.qm31_witness(cs, vals[0:4], ("perm", flow_idx, half, 0))
        r = C.qm31_witness(cs, vals[4:8], ("perm", flow_idx, half, 1))
        sel = cs.assemble_poseidon_gate(l.variable, r.variable)
        return Half(cs, vals, l.variable, r.variable, sel)

    new_left = result(out[0:8], ignore_left, 0)
    new_right = result(out[8:16], ignore_right, 1)
    cs.invoke_poseidon_accelerator((left.sel, left.value), (right.sel, right.value), (new_left.sel, new_left.value),
                                   (new_right.sel, new_right.value), is_swap[1] if is_swap else 0, bool(swapped))
    return new_left, new_right


def permute_get_rate(left, right):
    return permute(left, right, False, True)[0]


def permute_get_capacity(left, right):
    return permute(left, right, True, False)[1]


def swap_permute_get_rate(left, right, is_swap):
    return permute(left, right, False, True, is_swap)[0]


def half_equalverify(a: Half, b: Half):
    a.cs.insert_gate(a.left, 0, b.left, 1)
    a.cs.insert_gate(a.right, 0, b.right, 1)


# ---------------------------------------------------------------- channel
class Channel:
    def __init__(self, cs):
        self.cs = cs
        self.n_sent = 0
        self.digest = half_zero(cs)

    def mix_root(self, root: Half):
        self.digest = permute_get_capacity(root, self.digest)
        self.n_sent = 0

    def draw_felts(self):
        cs = self.cs
        n_sent = C.as_qm31(C.m31_constant(cs, self.n_sent))
        self.n_sent += 1
        left = half_from_qm31(n_sent, C.qm31_zero(cs))
        return permute_get_rate(left, self.digest).to_qm31()

    def mix_one_felt(self, felt: Var):
        left = half_from_qm31(felt, C.qm31_zero(self.cs))
        self.digest = permute_get_capacity(left, self.digest)
        self.n_sent = 0

    def mix_two_felts(self, f1: Var, f2: Var):
        left = half_from_qm31(f1, f2)
        self.digest = permute_get_capacity(left, self.digest)
        self.n_sent = 0


# ---------------------------------------------------------------- Merkle hasher
def hash_tree_with_swap(left, right, bit_value, bit_variable):
    return swap_permute_get_rate(left, right, (bit_value, bit_variable))


def hash_tree_with_column_hash_with_swap(left, right, bit_value, bit_variable, column_hash):
    h = swap_permute_get_rate(left, right, (bit_value, bit_variable))
    return permute_get_rate(h, column_hash)


def combine_hash_tree_with_column(hash_tree, hash_column):
    return permute_get_rate(hash_tree, hash_column)


def hash_m31_columns_get_capacity(m31):
    """merkle/src/lib.rs:166-208 (and the first part of hash_m31_columns_get_rate, :50-91)."""
    n = len(m31)
    cs = m31[0].cs
    num_chunk = -(-n // 8)
    inp = [C.m31_zero(cs) for _ in range(8)]
    inp[0:min(n, 8)] = m31[0:min(n, 8)]
    zero = half_zero(cs)
    first = half_from_m31(inp)
    digest = permute_get_capacity(first, zero)
    if num_chunk == 1:
        return digest
    for k in range(1, num_chunk - 1):
        digest = permute_get_capacity(half_from_m31(m31[8 * k:8 * k + 8]), digest)
    remain = n % 8
    inp = [C.m31_zero(cs) for _ in range(8)]
    if remain == 0:
        inp[0:8] = m31[n - 8:]
    else:
        inp[0:remain] = m31[n - remain:]
    return permute_get_capacity(half_from_m31(inp), digest)


def hash_m31_columns_get_rate(m31):
    cs = m31[0].cs
    digest = hash_m31_columns_get_capacity(m31)
    return permute_get_rate(half_zero(cs), digest)


def hash_qm31_columns_get_capacity(qm31):
    n = len(qm31)
    cs = qm31[0].cs
    num_chunk = -(-n // 2)
    inp = [C.qm31_zero(cs), C.qm31_zero(cs)]
    inp[0:min(n, 2)] = qm31[0:min(n, 2)]
    zero = half_zero(cs)
    first = half_from_qm31(inp[0], inp[1])
    digest = permute_get_capacity(first, zero)
    if num_chunk == 1:
        return digest
    for k in range(1, num_chunk - 1):
        digest = permute_get_capacity(half_from_qm31(qm31[2 * k], qm31[2 * k + 1]), digest)
    remain = n % 2
    inp = [C.qm31_zero(cs), C.qm31_zero(cs)]
    if remain == 0:
        inp[0:2] = qm31[n - 2:]
    else:
        inp[0:remain] = qm31[n - remain:]
    return permute_get_capacity(half_from_qm31(inp[0], inp[1]), digest)


def hash_qm31_columns_get_rate(qm31):
    cs = qm31[0].cs
    digest = hash_qm31_columns_get_capacity(qm31)
    return permute_get_rate(half_zero(cs), digest)


# ---------------------------------------------------------------- circle points
class PointM31:
    __slots__ = ("x", "y")

    def __init__(self, x: Var, y: Var):
        self.x, self.y = x, y


def pm_constant(cs, p):
    return PointM31(C.m31_constant(cs, p[0]), C.m31_constant(cs, p[1]))


def pm_add(a: PointM31, b: PointM31) -> PointM31:
    x1x2 = C.mul(a.x, b.x)
    y1y2 = C.mul(a.y, b.y)
    x1y2 = C.mul(a.x, b.y)
    y1x2 = C.mul(a.y, b.x)
    return PointM31(C.sub(x1x2, y1y2), C.add(x1y2, y1x2))


def pm_double(a: PointM31) -> PointM31:
    xx = C.mul(a.x, a.x)
    yy = C.mul(a.y, a.y)
    xy = C.mul(a.x, a.y)
    return PointM31(C.sub(xx, yy), C.mul_constant(xy, 2))


def pm_select(cs, point, bit_value, bit_variable) -> PointM31:
    # the gate constants are taken from the SELECTED value (circle/src/lib.rs:83-98), so this gate's `op` follows the
    # witness bit: 0 when the bit is 0, the step's coordinate when it is 1
    value = point if bit_value else (1, 0)
    new_x = cs.mul_constant(bit_variable, (value[0] - 1) % P, program_k=(point[0] - 1) % P)
    new_x = cs.add(new_x, 1)
    new_y = cs.mul_constant(bit_variable, value[1], program_k=point[1])
    return PointM31(Var(cs, value[0], new_x, 1), Var(cs, value[1], new_y, 1))


def pm_conditional_negate(a: PointM31, bit_value, bit_variable) -> PointM31:
    cs = a.x.cs
    y_value = (P - a.y.value) % P if bit_value else a.y.value
    mult = cs.mul_constant(bit_variable, P - 2)
    mult = cs.add(mult, 1)
    y_var = cs.mul(mult, a.y.variable)
    return PointM31(a.x, Var(cs, y_value, y_var, 1))


class PointQM31:
    __slots__ = ("x", "y")

    def __init__(self, x: Var, y: Var):
        self.x, self.y = x, y


def pq_witness(cs, p, tag_x=None, tag_y=None):
    return PointQM31(C.qm31_witness(cs, p[0], tag_x), C.qm31_witness(cs, p[1], tag_y))


def pq_from_t(t: Var) -> PointQM31:
    cs = t.cs
    t_doubled = C.add(t, t)
    t_squared = C.mul(t, t)
    t_squared_plus_1 = C.add(t_squared, C.m31_one(cs))
    inv = C.qm31_inv(t_squared_plus_1)
    one_minus = C.add(C.neg(t_squared), C.m31_one(cs))
    return PointQM31(C.mul(one_minus, inv), C.mul(t_doubled, inv))


def pq_repeated_double_x_only(p: PointQM31, log_size: int) -> Var:
    x = p.x
    for _ in range(log_size):
        sq = C.mul(x, x)
        x = C.sub(C.add(sq, sq), C.m31_one(x.cs))
    return x


def pq_add_const(p: PointQM31, c) -> PointQM31:
    """&CirclePointQM31Var + &CirclePoint<M31> (circle/src/lib.rs:246-260)."""
    x1x2 = C.mul_constant(p.x, c[0])
    y1y2 = C.mul_constant(p.y, c[1])
    x1y2 = C.mul_constant(p.x, c[1])
    y1x2 = C.mul_constant(p.y, c[0])
    return PointQM31(C.sub(x1x2, y1y2), C.add(x1y2, y1x2))


# ---------------------------------------------------------------- line polynomial
class LinePoly:
    def __init__(self, cs, coeffs):
        self.cs, self.coeffs = cs, coeffs


def line_poly_witness(cs, coeffs, tag=None):
    return LinePoly(cs, [C.qm31_witness(cs, c, tag and tag + (k,)) for k, c in enumerate(coeffs)])


def line_eval_at_point(poly: LinePoly, x: Var) -> Var:
    cs = poly.cs
    log_size = (len(poly.coeffs)).bit_length() - 1
    doublings = [x]
    for _ in range(1, log_size):
        x_sq = C.mul(x, x)
        x = C.add(x_sq, x_sq)
        x = C.add(x, C.m31_constant(cs, P - 1))
        doublings.append(x)

    def fold(values, factors):
        n = len(values)
        if n == 1:
            return values[0]
        lhs = fold(values[:n // 2], factors[1:])
        rhs = fold(values[n // 2:], factors[1:])
        return C.add(lhs, C.mul(rhs, factors[0]))

    return fold(poly.coeffs, doublings)


# ---------------------------------------------------------------- query positions
class PointCarryingQuery:
    def __init__(self, bits: Bits, last_step, point: PointM31):
        self.bits, self.last_step, self.point = bits, last_step, point

    def copy(self):
        return PointCarryingQuery(self.bits, self.last_step, self.point)

    def get_next_point(self) -> PointM31:
        return pm_conditional_negate(pm_double(self.point), self.bits.value[0], self.bits.variables[0])

    def get_next_point_x(self) -> Var:
        xx = C.mul(self.point.x, self.point.x)
        yy = C.mul(self.point.y, self.point.y)
        return C.sub(xx, yy)

    def next(self):
        cs = self.bits.cs
        t = pm_select(cs, self.last_step, self.bits.value[1], self.bits.variables[1])
        self.bits = self.bits.index_range_from(1)
        self.point = pm_double(pm_add(self.point, t))

    def get_absolute_point(self) -> PointM31:
        return self.point


def point_carrying_query(bits: Bits) -> PointCarryingQuery:
    """PointCarryingQueryVar::new (query/src/lib.rs:62-137)."""
    cs = bits.cs
    log_size = len(bits.value)
    coset = canonic_half_coset(log_size + 1)
    steps = []
    cur = coset.step
    for _ in range(log_size - 1):
        steps.append(cur)
        cur = cp_double(cur)
    combs = list(zip(steps, reversed(bits.value[1:]), reversed(bits.variables[1:])))
    cur = pm_constant(cs, coset.initial)
    for k in range(0, len(combs), 2):
        chunk = combs[k:k + 2]
        if len(chunk) == 1:
            point = pm_select(cs, chunk[0][0], chunk[0][1], chunk[0][2])
            cur = pm_add(point, cur)
        else:
            p00 = (1, 0)
            p01 = chunk[0][0]
            p10 = chunk[1][0]
            p11 = cp_add(p01, p10)
            value = {(False, False): p00, (True, False): p01, (False, True): p10, (True, True): p11}[(bool(chunk[0][1]), bool(chunk[1][1]))]
            a, b = chunk[0][2], chunk[1][2]
            one_minus_a = cs.add(1, cs.mul_constant(a, P - 1))
            one_minus_b = cs.add(1, cs.mul_constant(b, P - 1))
            b00 = cs.mul(one_minus_a, one_minus_b)
            b01 = cs.mul(a, one_minus_b)
            b10 = cs.mul(one_minus_a, b)
            b11 = cs.mul(a, b)
            x = cs.mul_constant(b00, p00[0])
            x = cs.add(x, cs.mul_constant(b01, p01[0]))
            x = cs.add(x, cs.mul_constant(b10, p10[0]))
            x = cs.add(x, cs.mul_constant(b11, p11[0]))
            y = cs.mul_constant(b00, p00[1])
            y = cs.add(y, cs.mul_constant(b01, p01[1]))
            y = cs.add(y, cs.mul_constant(b10, p10[1]))
            y = cs.add(y, cs.mul_constant(b11, p11[1]))
            point = PointM31(Var(cs, value[0], x, 1), Var(cs, value[1], y, 1))
            cur = pm_add(point, cur)
    return PointCarryingQuery(bits, cp_neg(steps[-1]), cur)


class QueryPositionsPerLogSize:
    """QueryPositionsPerLogSizeVar::new (query/src/lib.rs:19-48): points[log_size] = one PointCarryingQuery per raw query."""

    def __init__(self, min_degree, max_degree, raw_queries):
        elems = [point_carrying_query(bits_from_m31(q, 31).index_range(0, max_degree)) for q in raw_queries]
        self.points = {max_degree: [e.copy() for e in elems]}
        for log_size in range(max_degree - 1, min_degree - 1, -1):
            for e in elems:
                e.next()
            self.points[log_size] = [e.copy() for e in elems]

    def __getitem__(self, log_size):
        return self.points[log_size]
