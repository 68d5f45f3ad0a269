"""TEST INFRASTRUCTURE (oracle): the recursion circuit itself — proof allocation, Fiat-Shamir, composition check, DEEP answers + decommitments, FRI
folding — in the order examples/multi-proofs/src/main.rs:66-139 (and examples/single-proof/src/main.rs) runs them,
restated over cs.py / gadgets.py.

Reference: components/recursive/data_structures/src/lib.rs (the *Var allocations), fiat_shamir/src/lib.rs:31-180,
composition/src/{lib,plonk,poseidon,data_structures}.rs, answer/src/{lib,data_structures}.rs, folding/src/lib.rs:11-206.

One thing cannot be restated: AnswerResults::compute walks two std HashSet<isize> ({0, -1} each) to build the shifted
OODS points (answer/src/lib.rs:44-71), and Rust's HashSet order is seeded per process — the reference's own circuit
differs from run to run in the order of those two pairs of blocks.  `shift_order` picks one of the four (the default is
as good as any; tests/test_recursion_circuit.py finds the one each of the reference's fixtures was made with).
"""
from __future__ import annotations

from . import cs as C
from . import gadgets as G
from .cs import P, Var

PLONK_COLS = (10, 12, 8)
POSEIDON_COLS = (40, 48, 8)


# ---------------------------------------------------------------- PlonkWithPoseidonProofVar::new_witness
class ProofVar:
    pass


def allocate_proof(cs, d) -> ProofVar:
    """data_structures/src/lib.rs:36-46, 71-80, 108-119, 137-154, 172-219."""
    pv = ProofVar()
    pv.cs = cs
    pv.log_size_plonk = C.m31_witness(cs, d.lp, ("in", "lp"))
    pv.log_size_poseidon = C.m31_witness(cs, d.lq, ("in", "lq"))
    pv.plonk_total_sum = C.qm31_witness(cs, d.plonk_total_sum, ("in", "plonk_sum"))
    pv.poseidon_total_sum = C.qm31_witness(cs, d.poseidon_total_sum, ("in", "poseidon_sum"))
    pv.commitments = [G.half_witness(cs, d.commitments[t], ("in", "commit", t)) for t in range(4)]
    pv.sampled_values = [[[C.qm31_witness(cs, v, ("in", "sample", t, c, s)) for s, v in enumerate(col)] for c, col in enumerate(tree)]
                         for t, tree in enumerate(d.sampled_values)]
    pv.first_layer_commitment = G.half_witness(cs, d.first_layer_commitment, ("in", "fri_commit", 0))
    pv.inner_layer_commitments = [G.half_witness(cs, c, ("in", "fri_commit", 1 + k)) for k, c in enumerate(d.inner_layer_commitments)]
    pv.last_poly = G.line_poly_witness(cs, d.last_poly, ("in", "last_poly"))
    n = d.nonce
    pv.proof_of_work = [C.m31_witness(cs, n & ((1 << 22) - 1), ("in", "nonce", 0)),
                        C.m31_witness(cs, (n >> 22) & ((1 << 21) - 1), ("in", "nonce", 1)),
                        C.m31_witness(cs, (n >> 43) & ((1 << 21) - 1), ("in", "nonce", 2))]
    return pv


# ---------------------------------------------------------------- FiatShamirResults::compute
class FiatShamir:
    pass


def lookup_elements(z: Var, alpha: Var):
    """LookupElementsVar::from_z_and_alpha (data_structures/src/lib.rs:243-263)."""
    cs = z.cs
    powers = [C.qm31_one(cs), alpha]
    cur = alpha
    for _ in range(2, 3):
        cur = C.mul(cur, alpha)
        powers.append(cur)
    return {"z": z, "alpha": alpha, "alpha_powers": powers}


def fiat_shamir(pv: ProofVar, d, inputs) -> FiatShamir:
    cs = pv.cs
    fs = FiatShamir()
    ch = G.Channel(cs)
    ch.mix_root(pv.commitments[0])
    ch.mix_one_felt(C.as_qm31(pv.log_size_plonk))
    ch.mix_one_felt(C.as_qm31(pv.log_size_poseidon))
    ch.mix_root(pv.commitments[1])
    z, alpha = ch.draw_felts()
    fs.lookup = lookup_elements(z, alpha)
    ch.mix_two_felts(pv.plonk_total_sum, pv.poseidon_total_sum)
    ch.mix_root(pv.commitments[2])
    fs.random_coeff = ch.draw_felts()[0]
    ch.mix_root(pv.commitments[3])
    t = ch.draw_felts()[0]
    fs.oods_point = G.pq_from_t(t)
    flat = [v for tree in pv.sampled_values for col in tree for v in col]
    for k in range(0, len(flat), 2):
        if k + 1 == len(flat):
            ch.mix_one_felt(flat[k])
        else:
            ch.mix_two_felts(flat[k], flat[k + 1])
    fs.after_sampled_values_random_coeff = ch.draw_felts()[0]
    fs.fri_alphas = []
    ch.mix_root(pv.first_layer_commitment)
    fs.fri_alphas.append(ch.draw_felts()[0])
    for l in pv.inner_layer_commitments:
        ch.mix_root(l)
        fs.fri_alphas.append(ch.draw_felts()[0])
    coeffs = pv.last_poly.coeffs
    for k in range(0, len(coeffs), 2):
        if k + 1 == len(coeffs):
            ch.mix_one_felt(coeffs[k])
        else:
            ch.mix_two_felts(coeffs[k], coeffs[k + 1])
    nonce_felt = C.qm31_from_m31(pv.proof_of_work[0], pv.proof_of_work[1], pv.proof_of_work[2], C.m31_zero(cs))
    G.bits_from_m31(pv.proof_of_work[0], 22)
    G.bits_from_m31(pv.proof_of_work[1], 21)
    G.bits_from_m31(pv.proof_of_work[2], 21)
    ch.mix_one_felt(nonce_felt)
    lower_bits = G.bits_from_m31(C.decompose_m31(ch.digest.to_qm31()[0])[0], 31).compose_range(0, d.pow_bits)
    C.equalverify(lower_bits, C.m31_zero(cs))
    felts = []
    for _ in range(-(-d.nq // 4)):
        a, b = ch.draw_felts()
        felts.append(a)
        felts.append(b)
    raw_queries = []
    for felt in felts:
        raw_queries.extend(C.decompose_m31(felt))
    fs.raw_queries = raw_queries[:d.nq]
    input_sum = C.qm31_zero(cs)
    for idx, v in inputs:
        s = C.sub(C.add(v, C.mul(C.qm31_constant(cs, (idx, 0, 0, 0)), alpha)), z)
        input_sum = C.add(input_sum, C.qm31_inv(s))
    C.equalverify(C.add(C.add(input_sum, pv.poseidon_total_sum), pv.plonk_total_sum), C.qm31_zero(cs))
    return fs


# ---------------------------------------------------------------- CompositionCheck::compute
def coset_vanishing(p: G.PointQM31, coset_log_size: int) -> Var:
    cs = p.x.cs
    coset = G.canonic_coset(coset_log_size)
    shift = G.cp_add(G.cp_neg(coset.initial), G.point_of_index(coset.step_size >> 1))
    x = G.pq_add_const(p, shift).x
    for _ in range(1, coset.log_size):
        sq = C.mul(x, x)
        x = C.sub(C.add(sq, sq), C.m31_one(cs))
    return x


class Accumulator:
    def __init__(self, random_coeff):
        self.random_coeff = random_coeff
        self.accumulation = C.qm31_zero(random_coeff.cs)

    def accumulate(self, evaluation):
        self.accumulation = C.add(C.mul(self.accumulation, self.random_coeff), evaluation)


def combine_ef(v):
    return C.add(C.add(C.add(v[0], C.shift_by_i(v[1])), C.shift_by_j(v[2])), C.shift_by_ij(v[3]))


class EvalAtRow:
    """EvalAtRowVar + LogupAtRowVar (composition/src/data_structures.rs:57-215)."""

    def __init__(self, mask, total_sum, denom_inverse, log_size, acc):
        self.col_index = [0, 0, 0, 0]
        self.mask = mask
        self.cumsum_shift = C.mul_constant(total_sum, C.m_inv(pow(2, log_size, P)))
        self.fracs = []
        self.denom_inverse = denom_inverse
        self.acc = acc

    def next_mask(self, interaction):
        k = self.col_index[interaction]
        self.col_index[interaction] += 1
        return self.mask[interaction][k]

    def next_trace_mask(self):
        return self.next_mask(1)[0]

    def get_preprocessed_column(self):
        return self.next_mask(0)[0]

    def next_extension_interaction_mask(self, n):
        cols = [self.next_mask(2) for _ in range(4)]
        assert all(len(c) == n for c in cols)
        return [combine_ef([cols[k][s] for k in range(4)]) for s in range(n)]

    def add_to_relation(self, lk, multiplicity, values):
        denom = C.mul(lk["alpha_powers"][0], values[0])
        for ap, v in list(zip(lk["alpha_powers"], values))[1:]:
            denom = C.add(denom, C.mul(ap, v))
        denom = C.sub(denom, lk["z"])
        self.fracs.append((multiplicity, denom))

    def add_constraint(self, value):
        self.acc.accumulate(C.mul(value, self.denom_inverse))

    def finalize_logup(self, batch_size):
        cs = self.cumsum_shift.cs
        batched = []
        for k in range(0, len(self.fracs), batch_size):
            chunk = self.fracs[k:k + batch_size]
            if len(chunk) == 1:
                batched.append(chunk[0])
            else:
                p, q = chunk[0]
                for e in chunk[1:]:
                    p = C.add(C.mul(p, e[1]), C.mul(e[0], q))
                    q = C.mul(q, e[1])
                batched.append((p, q))
        prev_col_cumsum = C.qm31_zero(cs)
        for num, denom in batched[:-1]:
            cur, = self.next_extension_interaction_mask(1)
            diff = C.sub(cur, prev_col_cumsum)
            prev_col_cumsum = cur
            self.add_constraint(C.sub(C.mul(diff, denom), num))
        for num, denom in batched[-1:]:
            prev_row, cur = self.next_extension_interaction_mask(2)
            diff = C.sub(C.sub(cur, prev_row), prev_col_cumsum)
            fixed = C.add(diff, self.cumsum_shift)
            self.add_constraint(C.sub(C.mul(fixed, denom), num))


def evaluate_plonk(lk, ev: EvalAtRow):
    """composition/src/plonk.rs:8-82."""
    cs = lk["z"].cs
    a_wire, b_wire, c_wire, op = (ev.get_preprocessed_column() for _ in range(4))
    mult_a, mult_b, mult_c = (ev.get_preprocessed_column() for _ in range(3))
    poseidon_wire, mult_poseidon, enforce_c_m31 = (ev.get_preprocessed_column() for _ in range(3))
    a_v = [ev.next_trace_mask() for _ in range(4)]
    b_v = [ev.next_trace_mask() for _ in range(4)]
    c_v = [ev.next_trace_mask() for _ in range(4)]
    ev.add_constraint(C.mul(enforce_c_m31, c_v[1]))
    ev.add_constraint(C.mul(enforce_c_m31, c_v[2]))
    ev.add_constraint(C.mul(enforce_c_m31, c_v[3]))
    a_val, b_val, c_val = combine_ef(a_v), combine_ef(b_v), combine_ef(c_v)
    t1 = C.sub(c_val, C.mul(op, C.add(a_val, b_val)))
    t2 = C.mul(C.mul(C.sub(C.qm31_one(cs), op), a_val), b_val)
    ev.add_constraint(C.sub(t1, t2))
    ev.add_to_relation(lk, mult_a, [a_val, a_wire])
    ev.add_to_relation(lk, mult_b, [b_val, b_wire])
    ev.add_to_relation(lk, mult_c, [c_val, c_wire])
    ev.add_to_relation(lk, C.neg(mult_poseidon), [poseidon_wire, a_val, b_val])
    ev.finalize_logup(2)


def apply_m4(x):
    t0 = C.add(x[0], x[1])
    t02 = C.add(t0, t0)
    t1 = C.add(x[2], x[3])
    t12 = C.add(t1, t1)
    t2 = C.add(C.add(x[1], x[1]), t1)
    t3 = C.add(C.add(x[3], x[3]), t0)
    t4 = C.add(C.add(t12, t12), t3)
    t5 = C.add(C.add(t02, t02), t2)
    t6 = C.add(t3, t5)
    t7 = C.add(t2, t4)
    return [t6, t5, t7, t4]


def apply_external_round_matrix(state):
    for i in range(4):
        state[4 * i:4 * i + 4] = apply_m4(state[4 * i:4 * i + 4])
    for j in range(4):
        s = C.add(C.add(C.add(state[j], state[j + 4]), state[j + 8]), state[j + 12])
        for i in range(4):
            state[4 * i + j] = C.add(state[4 * i + j], s)


def apply_internal_round_matrix(state):
    total = state[0]
    for s in state[1:]:
        total = C.add(total, s)
    state[0] = C.add(state[0], C.add(C.add(state[0], state[0]), total))
    for i in range(1, 16):
        state[i] = C.add(C.mul_constant(state[i], 1 << (i + 1)), total)


def pow5(x):
    x2 = C.mul(x, x)
    x4 = C.mul(x2, x2)
    return C.mul(x4, x)


def evaluate_poseidon(lk, ev: EvalAtRow):
    """composition/src/poseidon.rs:73-241."""
    cs = lk["z"].cs
    is_first_round = ev.get_preprocessed_column()
    is_last_round = ev.get_preprocessed_column()
    is_full_round = ev.get_preprocessed_column()
    one = C.qm31_one(cs)
    is_not_first_round = C.sub(one, is_first_round)
    is_not_last_round = C.sub(one, is_last_round)
    is_partial_round = C.sub(is_not_first_round, is_full_round)
    round_id = ev.get_preprocessed_column()
    rc0 = [ev.get_preprocessed_column() for _ in range(16)]
    rc1 = [ev.get_preprocessed_column() for _ in range(16)]
    external_idx_1 = ev.get_preprocessed_column()
    external_idx_2 = ev.get_preprocessed_column()
    is_external_idx_1_nonzero = ev.get_preprocessed_column()
    is_external_idx_2_nonzero = ev.get_preprocessed_column()
    swap_bit_addr = rc0[0]
    in_state = [ev.next_trace_mask() for _ in range(16)]
    intermediate_state = [ev.next_trace_mask() for _ in range(16)]
    out_state = [ev.next_trace_mask() for _ in range(16)]
    swap_bit_value = intermediate_state[0]

    one_minus_swap = C.sub(one, swap_bit_value)
    permuted = []
    for i in range(16):
        if i < 8:
            permuted.append(C.add(C.mul(in_state[i], one_minus_swap), C.mul(in_state[i + 8], swap_bit_value)))
        else:
            permuted.append(C.add(C.mul(in_state[i - 8], swap_bit_value), C.mul(in_state[i], one_minus_swap)))
    apply_external_round_matrix(permuted)
    for i in range(16):
        ev.add_constraint(C.mul(is_first_round, C.sub(permuted[i], out_state[i])))

    full = list(in_state)
    for i in range(16):
        full[i] = C.add(full[i], rc0[i])
    full = [pow5(full[i]) for i in range(16)]
    for i in range(16):
        ev.add_constraint(C.mul(is_full_round, C.sub(intermediate_state[i], full[i])))
        full[i] = intermediate_state[i]
    apply_external_round_matrix(full)
    for i in range(16):
        full[i] = C.add(full[i], rc1[i])
    full = [pow5(full[i]) for i in range(16)]
    apply_external_round_matrix(full)
    for i in range(16):
        ev.add_constraint(C.mul(is_full_round, C.sub(out_state[i], full[i])))

    partial = list(in_state)
    for r in range(14):
        partial[0] = C.add(partial[0], rc0[r])
        partial[0] = pow5(partial[0])
        ev.add_constraint(C.mul(is_partial_round, C.sub(intermediate_state[r], partial[0])))
        partial[0] = intermediate_state[r]
        apply_internal_round_matrix(partial)
    for i in range(16):
        ev.add_constraint(C.mul(is_partial_round, C.sub(out_state[i], partial[i])))

    in_left_id = C.add(round_id, round_id)
    in_right_id = C.add(in_left_id, one)
    out_left_id = C.add(in_right_id, one)
    out_right_id = C.add(out_left_id, one)

    sel = C.mul(is_external_idx_1_nonzero, is_first_round)
    ident = C.add(C.mul(is_first_round, external_idx_1), C.mul(is_not_first_round, in_left_id))
    a = combine_ef(in_state[0:4])
    b = combine_ef(in_state[4:8])
    ev.add_to_relation(lk, C.sub(sel, is_not_first_round), [ident, a, b])

    sel = C.mul(is_external_idx_2_nonzero, is_first_round)
    ident = C.add(C.mul(is_first_round, external_idx_2), C.mul(is_not_first_round, in_right_id))
    a = combine_ef(in_state[8:12])
    b = combine_ef(in_state[12:16])
    ev.add_to_relation(lk, C.sub(sel, is_not_first_round), [ident, a, b])

    sel = C.mul(is_external_idx_1_nonzero, is_last_round)
    ident = C.add(C.mul(is_last_round, external_idx_1), C.mul(is_not_last_round, out_left_id))
    a = combine_ef(out_state[0:4])
    b = combine_ef(out_state[4:8])
    ev.add_to_relation(lk, C.add(sel, is_not_last_round), [ident, a, b])

    sel = C.mul(is_external_idx_2_nonzero, is_last_round)
    ident = C.add(C.mul(is_last_round, external_idx_2), C.mul(is_not_last_round, out_right_id))
    a = combine_ef(out_state[8:12])
    b = combine_ef(out_state[12:16])
    ev.add_to_relation(lk, C.add(sel, is_not_last_round), [ident, a, b])
    ev.add_to_relation(lk, C.mul(is_first_round, is_not_last_round), [swap_bit_value, swap_bit_addr])
    ev.finalize_logup(3)


def composition_check(pv: ProofVar, d, fs: FiatShamir):
    """composition/src/lib.rs:33-129."""
    acc = Accumulator(fs.random_coeff)
    sv = pv.sampled_values
    mask = [sv[0][0:10], sv[1][0:12], sv[2][0:8]]
    ev = EvalAtRow(mask, pv.plonk_total_sum, C.qm31_inv(coset_vanishing(fs.oods_point, d.lp)), d.lp, acc)
    evaluate_plonk(fs.lookup, ev)
    mask = [sv[0][10:50], sv[1][12:60], sv[2][8:16]]
    ev = EvalAtRow(mask, pv.poseidon_total_sum, C.qm31_inv(coset_vanishing(fs.oods_point, d.lq)), d.lq, acc)
    evaluate_poseidon(fs.lookup, ev)
    computed = acc.accumulation
    left = combine_ef([sv[3][k][0] for k in range(4)])
    right = combine_ef([sv[3][4 + k][0] for k in range(4)])
    comp_log_degree_bound = max(d.lp + 1, d.lq + 2) + 1
    expected = C.add(left, C.mul(right, G.pq_repeated_double_x_only(fs.oods_point, comp_log_degree_bound - 2)))
    C.equalverify(computed, expected)


# ---------------------------------------------------------------- SinglePath / SinglePair Merkle proofs
class PathProofVar:
    def __init__(self, cs, value, tag):
        """SinglePathMerkleProofVar::new (data_structures/src/lib.rs:283-310): sibling hashes take no variable; the
        column values are allocated in ascending order of their log size (BTreeMap)."""
        self.value = value
        self.sibling_hashes = [G.half_single_use(cs, h) for h in value.sibling_hashes]
        self.columns = {}
        for k in sorted(value.columns):
            self.columns[k] = [C.m31_witness(cs, v, tag + ("col", k, j)) for j, v in enumerate(value.columns[k])]

    def verify(self, root: G.Half, query: G.Bits):
        depth = self.value.depth
        assert query.get_value() == self.value.query % P
        cur = G.hash_m31_columns_get_rate(self.columns[depth])
        for i in range(depth):
            h = depth - i - 1
            if h in self.columns:
                column_hash = G.hash_m31_columns_get_capacity(self.columns[h])
                cur = G.hash_tree_with_column_hash_with_swap(cur, self.sibling_hashes[i], query.value[i], query.variables[i], column_hash)
            else:
                cur = G.hash_tree_with_swap(cur, self.sibling_hashes[i], query.value[i], query.variables[i])
        assert cur.value == root.value, "Merkle path does not reach the root"
        G.half_equalverify(cur, root)


class PairProofVar:
    def __init__(self, cs, value, tag):
        """SinglePairMerkleProofVar::new (data_structures/src/lib.rs:357-383)."""
        self.cs = cs
        self.value = value
        self.sibling_hashes = [G.half_single_use(cs, h) for h in value.sibling_hashes]
        self.self_columns = {k: C.qm31_witness(cs, value.self_columns[k], tag + ("self", k)) for k in sorted(value.self_columns)}
        self.siblings_columns = {k: C.qm31_witness(cs, value.siblings_columns[k], tag + ("sib", k)) for k in sorted(value.siblings_columns)}

    def verify(self, root: G.Half, query: G.Bits):
        cs = self.cs
        depth = self.value.depth
        assert query.get_value() == self.value.query % P
        self_hash = G.hash_qm31_columns_get_rate([self.self_columns[depth], C.qm31_zero(cs)])
        sibling_hash = G.hash_qm31_columns_get_rate([self.siblings_columns[depth], C.qm31_zero(cs)])
        for i in range(depth):
            h = depth - i - 1
            if h not in self.self_columns:
                self_hash = G.hash_tree_with_swap(self_hash, sibling_hash, query.value[i], query.variables[i])
                if i != depth - 1:
                    sibling_hash = self.sibling_hashes[i]
            else:
                self_column_hash = G.hash_qm31_columns_get_capacity([self.self_columns[h], C.qm31_zero(cs)])
                sibling_column_hash = G.hash_qm31_columns_get_capacity([self.siblings_columns[h], C.qm31_zero(cs)])
                self_hash = G.hash_tree_with_column_hash_with_swap(self_hash, sibling_hash, query.value[i], query.variables[i], self_column_hash)
                sibling_hash = G.combine_hash_tree_with_column(self.sibling_hashes[i], sibling_column_hash)
        assert self_hash.value == root.value, "Merkle pair path does not reach the root"
        G.half_equalverify(self_hash, root)


# ---------------------------------------------------------------- AnswerResults::compute
class Answer:
    pass


def complex_conjugate_line_coeffs(point: G.PointQM31, value: Var, alpha: Var):
    value0, value1 = C.decompose_cm31(value)
    y0, y1 = C.decompose_cm31(point.y)
    a = value1
    c = y1
    b = C.sub(C.mul(value0, y1), C.mul(value1, y0))
    return C.mul(alpha, a), C.mul(alpha, b), C.mul(alpha, c)


def fri_answers_for_log_size(samples, random_coeff, query_positions, queried_values):
    """answer/src/lib.rs:366-396 with answer/src/data_structures.rs:43-215.  samples[c] = [(shift key, point, value)]."""
    cs = random_coeff.cs
    batches = {}  # IndexMap: insertion ordered
    for column_index, col in enumerate(samples):
        for key, point, value in col:
            batches.setdefault(key, []).append((point, column_index, value))
    sample_batches = [(entries[0][0], [(c, v) for _, c, v in entries]) for entries in batches.values()]
    alpha = C.qm31_constant(cs, (0, 0, P - 2, 0))
    line_coeffs = []
    for point, cvs in sample_batches:
        per = []
        for _, sampled_value in cvs:
            per.append(complex_conjugate_line_coeffs(point, sampled_value, alpha))
            alpha = C.mul(alpha, random_coeff)
        line_coeffs.append(per)
    domain_points, evals = [], []
    for query_position, row in zip(query_positions, queried_values):
        domain_point = query_position.get_next_point()
        denominator_inverses = []
        for point, _ in sample_batches:
            prx, pix = C.decompose_cm31(point.x)
            pry, piy = C.decompose_cm31(point.y)
            a = C.mul(C.sub(prx, domain_point.x), piy)
            b = C.mul(C.sub(pry, domain_point.y), pix)
            denominator_inverses.append(C.cm31_inv(C.sub(a, b)))
        row_acc = C.qm31_zero(cs)
        for (point, cvs), coeffs, dinv in zip(sample_batches, line_coeffs, denominator_inverses):
            numerator = C.qm31_zero(cs)
            for (column_index, _), (a, b, c) in zip(cvs, coeffs):
                value = C.mul(row[column_index], c)
                linear_term = C.add(C.mul(a, domain_point.y), b)
                numerator = C.add(numerator, C.sub(value, linear_term))
            row_acc = C.add(row_acc, C.mul(numerator, dinv))
        evals.append(row_acc)
        domain_points.append(domain_point)
    return domain_points, evals


def answer(pv: ProofVar, d, fs: FiatShamir, shift_order=((0, -1), (0, -1))) -> Answer:
    cs = pv.cs
    ans = Answer()
    oods_point = G.pq_witness(cs, (fs.oods_point.x.value, fs.oods_point.y.value), ("copy", fs.oods_point.x.variable),
                              ("copy", fs.oods_point.y.variable))  # examples/multi-proofs/src/main.rs:108
    step_plonk = G.canonic_coset(d.lp).step
    step_poseidon = G.canonic_coset(d.lq).step
    shifted_plonk, shifted_poseidon = {}, {}
    for i in shift_order[0]:
        shifted_plonk[i] = G.pq_add_const(oods_point, G.cp_mul(step_plonk, i))
    for i in shift_order[1]:
        shifted_poseidon[i] = G.pq_add_const(oods_point, G.cp_mul(step_poseidon, i))

    def mask_of(t, n_cols, shifted, log_size):
        cols = []
        for c in range(n_cols):
            if t == 0:
                cols.append([("zero", oods_point)])
            elif t == 2 and c >= 4:
                cols.append([(("shift", -1, log_size), shifted[-1]), ("zero", shifted[0])])
            else:
                cols.append([("zero", shifted[0])])
        return cols

    sampled_points = [mask_of(t, PLONK_COLS[t], shifted_plonk, d.lp) + mask_of(t, POSEIDON_COLS[t], shifted_poseidon, d.lq) for t in range(3)]
    sampled_points.append([[("zero", oods_point)] for _ in range(8)])
    samples = [[[(key, point, value) for (key, point), value in zip(pts, vals)] for pts, vals in zip(tp, tv)]
               for tp, tv in zip(sampled_points, pv.sampled_values)]
    A, B, M = d.A, d.B, d.M
    qp = G.QueryPositionsPerLogSize(d.log_last + d.blowup + 1, M, fs.raw_queries)
    all_log_sizes = sorted({A, B, M})
    for ls in all_log_sizes:
        assert [q.bits.get_value() for q in qp[ls]] == [x >> (M - ls) for x in d.queries_M]
    # DecommitmentVar::new, then the four trees' paths (answer/src/lib.rs:212-262)
    dec = [[PathProofVar(cs, p, ("path", t, i)) for i, p in enumerate(d.decommit[t])] for t in range(4)]
    for t in range(3):
        for i, q in enumerate(qp[max(A, B)]):
            dec[t][i].verify(pv.commitments[t], q.bits)
    for i, q in enumerate(qp[M]):
        dec[3][i].verify(pv.commitments[3], q.bits)
    queried_values = {}
    for ls in all_log_sizes:
        rows = []
        for i in range(len(qp[ls])):
            v = []
            for t in range(4):
                v.extend(dec[t][i].columns.get(ls, []))
            rows.append(v)
        queried_values[ls] = rows
    col_sizes = [[A] * PLONK_COLS[t] + [B] * POSEIDON_COLS[t] for t in range(3)] + [[M] * 8]
    flat = [(ls, s) for tls, ts in zip(col_sizes, samples) for ls, s in zip(tls, ts)]
    ans.fri_answers, ans.domain_points = [], []
    for ls in sorted(all_log_sizes, reverse=True):
        group = [s for l, s in flat if l == ls]
        pts, evals = fri_answers_for_log_size(group, fs.after_sampled_values_random_coeff, qp[ls], queried_values[ls])
        ans.domain_points.append(pts)
        ans.fri_answers.append(evals)
    ans.qp = qp
    ans.all_log_sizes = all_log_sizes
    return ans


# ---------------------------------------------------------------- FoldingResults::compute
def folding(pv: ProofVar, d, fs: FiatShamir, ans: Answer):
    cs = pv.cs
    M = d.M
    qp = ans.qp
    proofs = []
    for i, p in enumerate(d.first_layer):
        proof = PairProofVar(cs, p, ("pair", 0, i))
        proof.verify(pv.first_layer_commitment, qp[M][i].bits)
        proofs.append(proof)
    for ls, answers in zip(reversed(ans.all_log_sizes), ans.fri_answers):
        for i, fri_answer in enumerate(answers):
            C.equalverify(proofs[i].self_columns[ls], fri_answer)
    folded_results = {}
    for ls in ans.all_log_sizes:
        per = []
        for proof, query in zip(proofs, qp[ls]):
            self_val, sibling_val = proof.self_columns[ls], proof.siblings_columns[ls]
            point = G.pm_double(query.get_absolute_point())
            y_inv = C.m31_inv(point.y)
            left, right = C.swap(self_val, sibling_val, query.bits.value[0], query.bits.variables[0])
            new_left = C.add(left, right)
            new_right = C.mul(C.sub(left, right), y_inv)
            per.append(C.add(new_left, C.mul(new_right, fs.fri_alphas[M - ls])))
        folded_results[ls] = per
    log_size = M
    folded = [C.qm31_zero(cs) for _ in range(len(qp[M]))]
    for i in range(d.n_inner):
        if log_size in folded_results:
            fri_alpha = fs.fri_alphas[i]
            fri_alpha = C.mul(fri_alpha, fri_alpha)
            folded = [C.add(C.mul(fri_alpha, v), b) for v, b in zip(folded, folded_results[log_size])]
        log_size -= 1
        new_folded = []
        for k, (folded_result, query, p) in enumerate(zip(folded, qp[log_size], d.inner_layers[log_size])):
            merkle_proof = PairProofVar(cs, p, ("pair", 1 + i, k))
            self_val, sibling_val = merkle_proof.self_columns[log_size], merkle_proof.siblings_columns[log_size]
            C.equalverify(folded_result, self_val)
            point = query.get_absolute_point()
            x_inv = C.m31_inv(point.x)
            left, right = C.swap(self_val, sibling_val, query.bits.value[0], query.bits.variables[0])
            new_left = C.add(left, right)
            new_right = C.mul(C.sub(left, right), x_inv)
            new_folded.append(C.add(new_left, C.mul(new_right, fs.fri_alphas[i + 1])))
            merkle_proof.verify(pv.inner_layer_commitments[i], query.bits)
        folded = new_folded
    for query, v in zip(qp[log_size], folded):
        if len(pv.last_poly.coeffs) == 1:
            C.equalverify(v, pv.last_poly.coeffs[0])
        else:
            x = query.get_next_point_x()
            C.equalverify(v, G.line_eval_at_point(pv.last_poly, x))


# ---------------------------------------------------------------- the whole circuit
def verify_in_circuit(cs, d, inputs, shift_order=((0, -1), (0, -1))):
    """One copy of the verifier (the body of the `multipliers` loop, examples/multi-proofs/src/main.rs:66-139)."""
    pv = allocate_proof(cs, d)
    marks = {"proof": (cs.num_plonk_rows(), len(cs.flow))}
    fs = fiat_shamir(pv, d, inputs)
    marks["fiat_shamir"] = (cs.num_plonk_rows(), len(cs.flow))
    composition_check(pv, d, fs)
    marks["composition"] = (cs.num_plonk_rows(), len(cs.flow))
    ans = answer(pv, d, fs, shift_order)
    marks["answer"] = (cs.num_plonk_rows(), len(cs.flow))
    folding(pv, d, fs, ans)
    marks["folding"] = (cs.num_plonk_rows(), len(cs.flow))
    return marks


def standard_inputs(cs):
    return [(1, C.qm31_one(cs)), (2, C.qm31_i(cs)), (3, C.qm31_j(cs))]
