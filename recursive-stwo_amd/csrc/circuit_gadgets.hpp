// circuit_gadgets.hpp — HOST code: the reference's circuit gadgets over circuit_cs.hpp, mirrored in allocation order.
// bits (primitives/bits/src/lib.rs), Poseidon2HalfVar (primitives/poseidon31/src/lib.rs), ChannelVar
// (primitives/channel/src/lib.rs), Poseidon31MerkleHasherVar (primitives/merkle/src/lib.rs), circle points
// (primitives/circle/src/lib.rs), LinePolyVar (primitives/line/src/lib.rs), query positions (primitives/query/src/lib.rs).
#pragma once
#include "circuit_cs.hpp"

namespace rsv::circuit {

// ---------------------------------------------------------------- the M31 circle group (stwo core/circle.rs; published)
struct Pt { uint32_t x, y; };
inline Pt cp_add(Pt a, Pt b) { return {h_sub(h_mul(a.x, b.x), h_mul(a.y, b.y)), h_add(h_mul(a.x, b.y), h_mul(a.y, b.x))}; }
inline Pt cp_double(Pt a) { return cp_add(a, a); }
inline Pt cp_neg(Pt a) { return {a.x, h_neg(a.y)}; }
inline Pt cp_mul(Pt a, uint32_t k) {  // k mod 2^31 (the group's order)
    k &= 0x7fffffffu;
    Pt r{1, 0};
    while (k) { if (k & 1) r = cp_add(r, a); a = cp_double(a); k >>= 1; }
    return r;
}
inline Pt point_of_index(uint32_t idx) { return cp_mul(Pt{2, 1268011823u}, idx); }
inline uint32_t subgroup_gen_index(uint32_t log_size) { return 1u << (31 - log_size); }
struct Coset {
    uint32_t initial_index, step_size, log_size;
    Pt initial, step;
};
inline Coset make_coset(uint32_t initial_index, uint32_t log_size) {
    Coset c{initial_index & 0x7fffffffu, subgroup_gen_index(log_size), log_size, {}, {}};
    c.initial = point_of_index(c.initial_index);
    c.step = point_of_index(c.step_size);
    return c;
}
inline Coset canonic_coset(uint32_t log_size) { return make_coset(subgroup_gen_index(log_size + 1), log_size); }           // Coset::odds
inline Coset canonic_half_coset(uint32_t log_size) { return make_coset(subgroup_gen_index(log_size + 1), log_size - 1); }  // half_odds(n - 1)

// ---------------------------------------------------------------- bits
struct Bits {
    ConstraintSystem* cs;
    std::vector<uint8_t> value;
    std::vector<uint32_t> variables;
    uint32_t get_value() const {
        uint64_t v = 0;
        for (size_t k = 0; k < value.size(); k++) v += (uint64_t)value[k] << k;
        return (uint32_t)(v % MP);
    }
    Bits range(size_t lo, size_t hi) const {
        return Bits{cs, std::vector<uint8_t>(value.begin() + lo, value.begin() + hi), std::vector<uint32_t>(variables.begin() + lo, variables.begin() + hi)};
    }
    Var compose_range(size_t lo, size_t hi) const {
        uint64_t total = value[lo];
        uint32_t var = variables[lo];
        for (size_t i = lo + 1, shift = 1; i < hi; i++, shift++) {
            if (value[i]) total += 1ull << shift;
            const uint32_t shifted = cs->mul_constant(variables[i], 1u << shift);
            var = cs->add(var, shifted);
        }
        return mk(cs, {(uint32_t)(total % MP), 0, 0, 0}, var, 1);
    }
};
inline Bits bits_from_m31(const Var& v, size_t l) {
    ConstraintSystem* cs = v.cs;
    Bits res{cs, {}, {}};
    for (size_t k = 0; k < l; k++) {  // BitsVar::new_variables, Witness mode (bits/src/lib.rs:24-46)
        const uint8_t b = (v.value[0] >> k) & 1u;
        cs->set_hint(mk_instr(W_BIT, v.variable, 0, (uint32_t)k));
        const uint32_t bit = cs->new_qm31({b, 0, 0, 0}, WITNESS);
        res.value.push_back(b);
        res.variables.push_back(bit);
        const Var minus_one = m31_constant(cs, MP - 1);
        const uint32_t bit_minus_one = cs->add(bit, minus_one.variable);
        cs->insert_gate(bit, bit_minus_one, 0, 0);
    }
    Var rec = mk(cs, {res.value[0], 0, 0, 0}, res.variables[0], 1);
    for (size_t i = 1; i < l; i++) {
        const Var term = mul_constant(mk(cs, {res.value[i], 0, 0, 0}, res.variables[i], 1), 1u << i);
        rec = add(rec, term);
    }
    equalverify(rec, v);
    if (l == 31) {
        uint32_t product = cs->mul(res.variables[0], res.variables[1]);
        for (size_t i = 2; i < l; i++) product = cs->mul(product, res.variables[i]);
        cs->enforce_zero(product);
    }
    return res;
}

// ---------------------------------------------------------------- Poseidon2HalfVar (native form)
using Hash8 = std::array<uint32_t, 8>;
struct Half {
    ConstraintSystem* cs;
    Hash8 value;
    uint32_t left, right, sel;
    std::array<Var, 2> to_qm31() const {
        return {mk(cs, {value[0], value[1], value[2], value[3]}, left, 4), mk(cs, {value[4], value[5], value[6], value[7]}, right, 4)};
    }
};
inline Half half_single_use(ConstraintSystem* cs, const Hash8& v) { return Half{cs, v, 0, 0, 0}; }
inline Half half_from_m31(const Var* v8) {
    ConstraintSystem* cs = v8[0].cs;
    const Var left = qm31_from_m31(v8[0], v8[1], v8[2], v8[3]);
    const Var right = qm31_from_m31(v8[4], v8[5], v8[6], v8[7]);
    const uint32_t sel = cs->assemble_poseidon_gate(left.variable, right.variable);
    Hash8 h;
    for (int i = 0; i < 8; i++) h[i] = v8[i].value[0];
    return Half{cs, h, left.variable, right.variable, sel};
}
inline Half half_from_qm31(const Var& a, const Var& b) {
    ConstraintSystem* cs = a.cs;
    const uint32_t sel = cs->assemble_poseidon_gate(a.variable, b.variable);
    return Half{cs, {a.value[0], a.value[1], a.value[2], a.value[3], b.value[0], b.value[1], b.value[2], b.value[3]}, a.variable, b.variable, sel};
}
inline Half half_witness(ConstraintSystem* cs, const Hash8& v, const Instr& hint_left, const Instr& hint_right) {
    const Var left = qm31_witness(cs, {v[0], v[1], v[2], v[3]}, hint_left);
    const Var right = qm31_witness(cs, {v[4], v[5], v[6], v[7]}, hint_right);
    const uint32_t sel = cs->assemble_poseidon_gate(left.variable, right.variable);
    return Half{cs, v, left.variable, right.variable, sel};
}
inline Half half_zero(ConstraintSystem* cs) {
    auto it = cs->cache.find("poseidon2 zero_half");
    if (it == cs->cache.end()) it = cs->cache.emplace("poseidon2 zero_half", cs->assemble_poseidon_gate(0, 0)).first;
    return Half{cs, Hash8{}, 0, 0, it->second};
}

// The permutation's outputs are not computed here: they are the GPU's own PoseidonFlow records of the template proof,
// consumed in invocation order — and refused if a record's inputs are not the state the circuit is about to permute.
struct FlowSource {
    const uint32_t* rec = nullptr;  // [count][32]
    const uint8_t* swap = nullptr;  // [count]
    size_t count = 0, cursor = 0;
};

struct Gadgets {
    ConstraintSystem* cs;
    FlowSource* flow;

    // Poseidon2HalfVar::permute (poseidon31/src/lib.rs:282-423)
    std::pair<Half, Half> permute(const Half& left, const Half& right, bool ignore_left, bool ignore_right, bool have_swap, bool swap_value,
                                  uint32_t swap_variable) {
        const size_t k = flow->cursor % flow->count;
        const uint32_t* rec = flow->rec + 32 * k;
        const bool swapped = have_swap && swap_value;
        for (int i = 0; i < 8; i++)
            if (rec[i] != left.value[i] || rec[8 + i] != right.value[i]) throw std::runtime_error("PoseidonFlow record is not the circuit's invocation");
        if ((flow->swap[k] != 0) != swapped) throw std::runtime_error("PoseidonFlow swap bit is not the circuit's");
        const uint32_t flow_idx = (uint32_t)k;
        flow->cursor++;
        auto result = [&](const uint32_t* vals, bool ignore, uint32_t half) {
            Hash8 h;
            for (int i = 0; i < 8; i++) h[i] = vals[i];
            if (ignore) return Half{cs, h, 0, 0, 0};
            const Var l = qm31_witness(cs, {h[0], h[1], h[2], h[3]}, mk_instr(W_FLOW, 0, 0, flow_idx, 16 + 8 * half));
            const Var r = qm31_witness(cs, {h[4], h[5], h[6], h[7]}, mk_instr(W_FLOW, 0, 0, flow_idx, 16 + 8 * half + 4));
            const uint32_t sel = cs->assemble_poseidon_gate(l.variable, r.variable);
            return Half{cs, h, l.variable, r.variable, sel};
        };
        const Half new_left = result(rec + 16, ignore_left, 0);
        const Half new_right = result(rec + 24, ignore_right, 1);
        cs->flow.push_back(FlowRecord{{left.sel, right.sel, new_left.sel, new_right.sel}, have_swap ? swap_variable : 0u});
        return {new_left, new_right};
    }
    Half permute_get_rate(const Half& l, const Half& r) { return permute(l, r, false, true, false, false, 0).first; }
    Half permute_get_capacity(const Half& l, const Half& r) { return permute(l, r, true, false, false, false, 0).second; }
    Half swap_permute_get_rate(const Half& l, const Half& r, bool bit_value, uint32_t bit_variable) {
        return permute(l, r, false, true, true, bit_value, bit_variable).first;
    }
    void half_equalverify(const Half& a, const Half& b) {
        cs->insert_gate(a.left, 0, b.left, 1);
        cs->insert_gate(a.right, 0, b.right, 1);
    }

    // ---- Merkle hasher (merkle/src/lib.rs)
    Half hash_tree_with_swap(const Half& l, const Half& r, bool bit_value, uint32_t bit_variable) { return swap_permute_get_rate(l, r, bit_value, bit_variable); }
    Half hash_tree_with_column_hash_with_swap(const Half& l, const Half& r, bool bit_value, uint32_t bit_variable, const Half& column_hash) {
        const Half h = swap_permute_get_rate(l, r, bit_value, bit_variable);
        return permute_get_rate(h, column_hash);
    }
    Half combine_hash_tree_with_column(const Half& tree, const Half& column) { return permute_get_rate(tree, column); }
    Half hash_m31_columns_get_capacity(const std::vector<Var>& m31) {  // :166-208
        const size_t n = m31.size(), num_chunk = (n + 7) / 8;
        std::array<Var, 8> inp;
        for (size_t i = 0; i < 8; i++) inp[i] = i < n ? m31[i] : m31_zero(cs);
        const Half zero = half_zero(cs);
        const Half first = half_from_m31(inp.data());
        Half digest = permute_get_capacity(first, zero);
        if (num_chunk == 1) return digest;
        for (size_t k = 1; k + 1 < num_chunk; k++) {
            const Half left = half_from_m31(&m31[8 * k]);
            digest = permute_get_capacity(left, digest);
        }
        const size_t remain = n % 8, take = remain ? remain : 8;
        for (size_t i = 0; i < 8; i++) inp[i] = i < take ? m31[n - take + i] : m31_zero(cs);
        const Half left = half_from_m31(inp.data());
        return permute_get_capacity(left, digest);
    }
    Half hash_m31_columns_get_rate(const std::vector<Var>& m31) {
        const Half digest = hash_m31_columns_get_capacity(m31);
        const Half zero = half_zero(cs);
        return permute_get_rate(zero, digest);
    }
    // the circuit only ever hashes [value, 0] (data_structures/src/lib.rs:410-421): one chunk
    Half hash_qm31_pair_get_capacity(const Var& a, const Var& b) {
        const Half zero = half_zero(cs);
        const Half first = half_from_qm31(a, b);
        return permute_get_capacity(first, zero);
    }
    Half hash_qm31_pair_get_rate(const Var& a, const Var& b) {
        const Half digest = hash_qm31_pair_get_capacity(a, b);
        const Half zero = half_zero(cs);
        return permute_get_rate(zero, digest);
    }
};

// ---------------------------------------------------------------- channel (channel/src/lib.rs)
struct Channel {
    Gadgets* g;
    uint32_t n_sent = 0;
    Half digest;
    explicit Channel(Gadgets* g_) : g(g_), digest(half_zero(g_->cs)) {}
    void mix_root(const Half& root) { digest = g->permute_get_capacity(root, digest); n_sent = 0; }
    std::array<Var, 2> draw_felts() {
        ConstraintSystem* cs = g->cs;
        const Var ns = as_qm31(m31_constant(cs, n_sent));
        n_sent++;
        const Half left = half_from_qm31(ns, qm31_zero(cs));
        return g->permute_get_rate(left, digest).to_qm31();
    }
    void mix_one_felt(const Var& felt) {
        const Half left = half_from_qm31(felt, qm31_zero(g->cs));
        digest = g->permute_get_capacity(left, digest);
        n_sent = 0;
    }
    void mix_two_felts(const Var& f1, const Var& f2) {
        const Half left = half_from_qm31(f1, f2);
        digest = g->permute_get_capacity(left, digest);
        n_sent = 0;
    }
};

// ---------------------------------------------------------------- circle points (circle/src/lib.rs)
struct PointM31 { Var x, y; };
inline PointM31 pm_constant(ConstraintSystem* cs, Pt p) {
    const Var x = m31_constant(cs, p.x);
    const Var y = m31_constant(cs, p.y);
    return {x, y};
}
inline PointM31 pm_add(const PointM31& a, const PointM31& b) {
    const Var x1x2 = mul(a.x, b.x);
    const Var y1y2 = mul(a.y, b.y);
    const Var x1y2 = mul(a.x, b.y);
    const Var y1x2 = mul(a.y, b.x);
    const Var nx = sub(x1x2, y1y2);
    const Var ny = add(x1y2, y1x2);
    return {nx, ny};
}
inline PointM31 pm_double(const PointM31& a) {
    const Var xx = mul(a.x, a.x);
    const Var yy = mul(a.y, a.y);
    const Var xy = mul(a.x, a.y);
    const Var nx = sub(xx, yy);
    const Var ny = mul_constant(xy, 2);
    return {nx, ny};
}
// the gate constants are taken from the SELECTED value (circle/src/lib.rs:83-98), so the reference's gate follows the
// witness bit; the program multiplies by the step's coordinate for every proof — the same product
inline PointM31 pm_select(ConstraintSystem* cs, Pt point, bool bit_value, uint32_t bit_variable) {
    const Pt value = bit_value ? point : Pt{1, 0};
    uint32_t nx = cs->mul_constant(bit_variable, h_sub(value.x, 1), h_sub(point.x, 1));
    nx = cs->add(nx, 1);
    const uint32_t ny = cs->mul_constant(bit_variable, value.y, point.y);
    return {mk(cs, {value.x, 0, 0, 0}, nx, 1), mk(cs, {value.y, 0, 0, 0}, ny, 1)};
}
inline PointM31 pm_conditional_negate(const PointM31& a, bool bit_value, uint32_t bit_variable) {
    ConstraintSystem* cs = a.x.cs;
    uint32_t m = cs->mul_constant(bit_variable, MP - 2);
    m = cs->add(m, 1);
    const uint32_t yv = cs->mul(m, a.y.variable);
    return {a.x, mk(cs, {bit_value ? h_neg(a.y.value[0]) : a.y.value[0], 0, 0, 0}, yv, 1)};
}
struct PointQM31 { Var x, y; };
inline PointQM31 pq_from_t(const Var& t) {
    ConstraintSystem* cs = t.cs;
    const Var t_doubled = add(t, t);
    const Var t_squared = mul(t, t);
    const Var plus1 = add(t_squared, m31_one(cs));
    const Var inv = qm31_inv(plus1);
    const Var one_minus = add(neg(t_squared), m31_one(cs));
    const Var x = mul(one_minus, inv);
    const Var y = mul(t_doubled, inv);
    return {x, y};
}
inline Var pq_repeated_double_x_only(const PointQM31& p, uint32_t log_size) {
    Var x = p.x;
    for (uint32_t i = 0; i < log_size; i++) {
        const Var sq = mul(x, x);
        const Var dbl = add(sq, sq);
        x = sub(dbl, m31_one(x.cs));
    }
    return x;
}
inline PointQM31 pq_add_const(const PointQM31& p, Pt c) {  // &CirclePointQM31Var + &CirclePoint<M31> (:246-260)
    const Var x1x2 = mul_constant(p.x, c.x);
    const Var y1y2 = mul_constant(p.y, c.y);
    const Var x1y2 = mul_constant(p.x, c.y);
    const Var y1x2 = mul_constant(p.y, c.x);
    const Var nx = sub(x1x2, y1y2);
    const Var ny = add(x1y2, y1x2);
    return {nx, ny};
}

// ---------------------------------------------------------------- line polynomial (line/src/lib.rs)
inline Var line_fold(const Var* values, size_t n, const Var* factors) {
    if (n == 1) return values[0];
    const Var lhs = line_fold(values, n / 2, factors + 1);
    const Var rhs = line_fold(values + n / 2, n / 2, factors + 1);
    return add(lhs, mul(rhs, factors[0]));
}
inline Var line_eval_at_point(ConstraintSystem* cs, const std::vector<Var>& coeffs, Var x) {
    uint32_t log_size = 0;
    while ((2u << log_size) <= coeffs.size()) log_size++;
    std::vector<Var> doublings{x};
    for (uint32_t i = 1; i < log_size; i++) {
        const Var x_sq = mul(x, x);
        x = add(x_sq, x_sq);
        x = add(x, m31_constant(cs, MP - 1));
        doublings.push_back(x);
    }
    return line_fold(coeffs.data(), coeffs.size(), doublings.data());
}

// ---------------------------------------------------------------- query positions (query/src/lib.rs)
struct PointCarryingQuery {
    Bits bits;
    Pt last_step;
    PointM31 point;
    PointM31 get_next_point() const { return pm_conditional_negate(pm_double(point), bits.value[0], bits.variables[0]); }
    Var get_next_point_x() const {
        const Var xx = mul(point.x, point.x);
        const Var yy = mul(point.y, point.y);
        return sub(xx, yy);
    }
    void next() {
        const PointM31 t = pm_select(bits.cs, last_step, bits.value[1], bits.variables[1]);
        bits = bits.range(1, bits.value.size());
        point = pm_double(pm_add(point, t));
    }
};
inline PointCarryingQuery point_carrying_query(const Bits& bits) {  // PointCarryingQueryVar::new (:62-137)
    ConstraintSystem* cs = bits.cs;
    const uint32_t log_size = (uint32_t)bits.value.size();
    const Coset coset = canonic_half_coset(log_size + 1);
    std::vector<Pt> steps;
    Pt cur_step = coset.step;
    for (uint32_t i = 0; i + 1 < log_size; i++) { steps.push_back(cur_step); cur_step = cp_double(cur_step); }
    // steps zipped with the bits above bit 0, highest first
    struct Comb { Pt step; bool value; uint32_t variable; };
    std::vector<Comb> combs;
    for (size_t k = 0; k < steps.size(); k++) combs.push_back({steps[k], bits.value[log_size - 1 - k] != 0, bits.variables[log_size - 1 - k]});
    PointM31 cur = pm_constant(cs, coset.initial);
    for (size_t k = 0; k < combs.size(); k += 2) {
        if (k + 1 == combs.size()) {
            const PointM31 point = pm_select(cs, combs[k].step, combs[k].value, combs[k].variable);
            cur = pm_add(point, cur);
        } else {
            const Pt p00{1, 0}, p01 = combs[k].step, p10 = combs[k + 1].step, p11 = cp_add(p01, p10);
            const Pt value = combs[k].value ? (combs[k + 1].value ? p11 : p01) : (combs[k + 1].value ? p10 : p00);
            const uint32_t a = combs[k].variable, b = combs[k + 1].variable;
            uint32_t t = cs->mul_constant(a, MP - 1);
            const uint32_t one_minus_a = cs->add(1, t);
            t = cs->mul_constant(b, MP - 1);
            const uint32_t one_minus_b = cs->add(1, t);
            const uint32_t b00 = cs->mul(one_minus_a, one_minus_b);
            const uint32_t b01 = cs->mul(a, one_minus_b);
            const uint32_t b10 = cs->mul(one_minus_a, b);
            const uint32_t b11 = cs->mul(a, b);
            uint32_t x = cs->mul_constant(b00, p00.x);
            t = cs->mul_constant(b01, p01.x); x = cs->add(x, t);
            t = cs->mul_constant(b10, p10.x); x = cs->add(x, t);
            t = cs->mul_constant(b11, p11.x); x = cs->add(x, t);
            uint32_t y = cs->mul_constant(b00, p00.y);
            t = cs->mul_constant(b01, p01.y); y = cs->add(y, t);
            t = cs->mul_constant(b10, p10.y); y = cs->add(y, t);
            t = cs->mul_constant(b11, p11.y); y = cs->add(y, t);
            const PointM31 point{mk(cs, {value.x, 0, 0, 0}, x, 1), mk(cs, {value.y, 0, 0, 0}, y, 1)};
            cur = pm_add(point, cur);
        }
    }
    return PointCarryingQuery{bits, cp_neg(steps.back()), cur};
}

}  // namespace rsv::circuit
