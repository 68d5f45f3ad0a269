// circuit_verifier.hpp — HOST code: the recursion circuit itself — proof allocation, Fiat-Shamir, composition check, DEEP
// answers + decommitments, FRI folding — in the order examples/multi-proofs/src/main.rs:66-139 runs them, over
// circuit_cs.hpp / circuit_gadgets.hpp.  Reference: components/recursive/data_structures/src/lib.rs (the *Var
// allocations), fiat_shamir/src/lib.rs:31-180, composition/src/{lib,plonk,poseidon,data_structures}.rs,
// answer/src/{lib,data_structures}.rs, folding/src/lib.rs:11-206.
//
// AnswerResults::compute walks two std HashSet<isize> = {0, -1} to build the shifted OODS points (answer/src/lib.rs:44-71):
// Rust seeds them per process, so the reference's own circuit differs from run to run in the order of two pairs of blocks.
// The builder takes 0 before -1 for both unless told otherwise per copy (`walk`: bit 0 = the Plonk set is walked -1 first,
// bit 1 = the Poseidon set) — any order is one the reference produces; a given run of the reference used one of the four.
#pragma once
#include <algorithm>
#include <map>
#include <set>

#include "circuit_gadgets.hpp"

namespace rsv::circuit {

constexpr uint32_t PLONK_COLS_[3] = {10, 12, 8}, POSEIDON_COLS_[3] = {40, 48, 8};

// The template proof as the circuit sees it + the hint structs of components/hints, from the buffers of rsv_hints_out.
struct Template {
    uint32_t lp, lq, pow_bits, blowup, log_last, nq, n_inner, A, B, M;
    const uint32_t* w = nullptr;  // the proof's words
    uint32_t first_commit_off = 0, inner_commit_off[MAX_INNER] = {}, last_off = 0, last_n = 0, nonce_off = 0;
    std::vector<uint32_t> sample_off;  // word offset of every sampled value, flattened tree / column / sample
    std::vector<std::vector<uint32_t>> samples_per_col[4];  // [tree][col] -> indices into sample_off
    // hints (one proof): trace_sib [4][nq][M][8], trace_pos [4][nq], trace_cols [4][nq][64], fri_sib [1+n_inner][nq][M][8],
    // fri_cols [1+n_inner][nq][3][8]
    const uint32_t *trace_sib, *trace_pos, *trace_cols, *fri_sib, *fri_cols;

    std::vector<std::pair<uint32_t, uint32_t>> tree_levels(int t) const {  // (log size, columns), leaf level first
        if (t == 3) return {{M, 8}};
        if (A == B) return {{A, PLONK_COLS_[t] + POSEIDON_COLS_[t]}};
        std::vector<std::pair<uint32_t, uint32_t>> v{{A, PLONK_COLS_[t]}, {B, POSEIDON_COLS_[t]}};
        if (v[0].first < v[1].first) std::swap(v[0], v[1]);
        return v;
    }
    std::vector<uint32_t> all_log_sizes() const {  // ascending, distinct
        std::set<uint32_t> s{A, B, M};
        return {s.begin(), s.end()};
    }
};

struct ProofVar {
    Var log_size_plonk, log_size_poseidon, plonk_total_sum, poseidon_total_sum;
    std::vector<Half> commitments;
    std::vector<std::vector<std::vector<Var>>> sampled_values;
    Half first_layer_commitment;
    std::vector<Half> inner_layer_commitments;
    std::vector<Var> last_poly;
    std::array<Var, 3> proof_of_work;
};

inline Q4 words4(const uint32_t* w) { return {w[0], w[1], w[2], w[3]}; }
inline Hash8 words8(const uint32_t* w) { return {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]}; }

// PlonkWithPoseidonProofVar::new_witness (data_structures/src/lib.rs:36-46, 71-80, 108-119, 137-154, 172-219)
inline ProofVar allocate_proof(ConstraintSystem* cs, const Template& d) {
    ProofVar pv;
    const uint32_t* w = d.w;
    pv.log_size_plonk = m31_witness(cs, d.lp, mk_instr(W_WORD, 0, 0, W_LP));
    pv.log_size_poseidon = m31_witness(cs, d.lq, mk_instr(W_WORD, 0, 0, W_LQ));
    pv.plonk_total_sum = qm31_witness(cs, words4(w + W_PLONK_SUM), mk_instr(W_WORD4, 0, 0, W_PLONK_SUM));
    pv.poseidon_total_sum = qm31_witness(cs, words4(w + W_POSEIDON_SUM), mk_instr(W_WORD4, 0, 0, W_POSEIDON_SUM));
    for (uint32_t t = 0; t < 4; t++)
        pv.commitments.push_back(half_witness(cs, words8(w + W_COMMIT0 + 8 * t), mk_instr(W_WORD4, 0, 0, W_COMMIT0 + 8 * t),
                                              mk_instr(W_WORD4, 0, 0, W_COMMIT0 + 8 * t + 4)));
    pv.sampled_values.resize(4);
    for (int t = 0; t < 4; t++)
        for (const auto& col : d.samples_per_col[t]) {
            std::vector<Var> vals;
            for (uint32_t k : col) vals.push_back(qm31_witness(cs, words4(w + d.sample_off[k]), mk_instr(W_WORD4, 0, 0, d.sample_off[k])));
            pv.sampled_values[t].push_back(vals);
        }
    pv.first_layer_commitment = half_witness(cs, words8(w + d.first_commit_off), mk_instr(W_FRI_COMMIT, 0, 0, 0, 0), mk_instr(W_FRI_COMMIT, 0, 0, 0, 1));
    for (uint32_t l = 0; l < d.n_inner; l++)
        pv.inner_layer_commitments.push_back(half_witness(cs, words8(w + d.inner_commit_off[l]), mk_instr(W_FRI_COMMIT, 0, 0, 1 + l, 0),
                                                          mk_instr(W_FRI_COMMIT, 0, 0, 1 + l, 1)));
    for (uint32_t k = 0; k < d.last_n; k++) pv.last_poly.push_back(qm31_witness(cs, words4(w + d.last_off + 4 * k), mk_instr(W_LAST_POLY, 0, 0, k)));
    const uint64_t nonce = (uint64_t)w[d.nonce_off] | ((uint64_t)w[d.nonce_off + 1] << 32);
    pv.proof_of_work = {m31_witness(cs, (uint32_t)(nonce & ((1u << 22) - 1)), mk_instr(W_NONCE, 0, 0, 0)),
                        m31_witness(cs, (uint32_t)((nonce >> 22) & ((1u << 21) - 1)), mk_instr(W_NONCE, 0, 0, 1)),
                        m31_witness(cs, (uint32_t)((nonce >> 43) & ((1u << 21) - 1)), mk_instr(W_NONCE, 0, 0, 2))};
    return pv;
}

struct Lookup { Var z, alpha, alpha_powers[3]; };
struct FiatShamir {
    Lookup lookup;
    Var random_coeff, after_sampled_values_random_coeff;
    PointQM31 oods_point;
    std::vector<Var> raw_queries, fri_alphas;
};

// FiatShamirResults::compute (fiat_shamir/src/lib.rs:31-180)
inline FiatShamir fiat_shamir(Gadgets& g, ProofVar& pv, const Template& d, const std::vector<std::pair<uint32_t, Var>>& inputs) {
    ConstraintSystem* cs = g.cs;
    FiatShamir fs;
    Channel ch(&g);
    ch.mix_root(pv.commitments[0]);
    ch.mix_one_felt(as_qm31(pv.log_size_plonk));
    ch.mix_one_felt(as_qm31(pv.log_size_poseidon));
    ch.mix_root(pv.commitments[1]);
    {
        const auto za = ch.draw_felts();
        fs.lookup.z = za[0];
        fs.lookup.alpha = za[1];
        fs.lookup.alpha_powers[0] = qm31_one(cs);
        fs.lookup.alpha_powers[1] = za[1];
        fs.lookup.alpha_powers[2] = mul(za[1], za[1]);
    }
    ch.mix_two_felts(pv.plonk_total_sum, pv.poseidon_total_sum);
    ch.mix_root(pv.commitments[2]);
    fs.random_coeff = ch.draw_felts()[0];
    ch.mix_root(pv.commitments[3]);
    const Var t = ch.draw_felts()[0];
    fs.oods_point = pq_from_t(t);
    std::vector<Var> flat;
    for (const auto& tree : pv.sampled_values)
        for (const auto& col : tree)
            for (const Var& v : col) flat.push_back(v);
    for (size_t k = 0; k < flat.size(); k += 2) {
        if (k + 1 == flat.size()) ch.mix_one_felt(flat[k]);
        else ch.mix_two_felts(flat[k], flat[k + 1]);
    }
    fs.after_sampled_values_random_coeff = ch.draw_felts()[0];
    ch.mix_root(pv.first_layer_commitment);
    fs.fri_alphas.push_back(ch.draw_felts()[0]);
    for (const Half& l : pv.inner_layer_commitments) {
        ch.mix_root(l);
        fs.fri_alphas.push_back(ch.draw_felts()[0]);
    }
    for (size_t k = 0; k < pv.last_poly.size(); k += 2) {
        if (k + 1 == pv.last_poly.size()) ch.mix_one_felt(pv.last_poly[k]);
        else ch.mix_two_felts(pv.last_poly[k], pv.last_poly[k + 1]);
    }
    const Var nonce_felt = qm31_from_m31(pv.proof_of_work[0], pv.proof_of_work[1], pv.proof_of_work[2], m31_zero(cs));
    bits_from_m31(pv.proof_of_work[0], 22);
    bits_from_m31(pv.proof_of_work[1], 21);
    bits_from_m31(pv.proof_of_work[2], 21);
    ch.mix_one_felt(nonce_felt);
    {
        const Var first = decompose_m31(ch.digest.to_qm31()[0])[0];
        const Var lower_bits = bits_from_m31(first, 31).compose_range(0, d.pow_bits);
        equalverify(lower_bits, m31_zero(cs));
    }
    std::vector<Var> felts;
    for (uint32_t k = 0; k < (d.nq + 3) / 4; k++) {
        const auto ab = ch.draw_felts();
        felts.push_back(ab[0]);
        felts.push_back(ab[1]);
    }
    for (const Var& felt : felts) {
        const auto parts = decompose_m31(felt);
        for (const Var& p : parts) fs.raw_queries.push_back(p);
    }
    fs.raw_queries.resize(d.nq);
    Var input_sum = qm31_zero(cs);
    for (const auto& [idx, v] : inputs) {
        const Var c = qm31_constant(cs, {idx, 0, 0, 0});
        const Var ca = mul(c, fs.lookup.alpha);
        const Var s = sub(add(v, ca), fs.lookup.z);
        const Var inv = qm31_inv(s);
        input_sum = add(input_sum, inv);
    }
    const Var total = add(add(input_sum, pv.poseidon_total_sum), pv.plonk_total_sum);
    equalverify(total, qm31_zero(cs));
    return fs;
}

// ---------------------------------------------------------------- CompositionCheck::compute
inline Var coset_vanishing(const PointQM31& p, uint32_t coset_log_size) {
    ConstraintSystem* cs = p.x.cs;
    const Coset coset = canonic_coset(coset_log_size);
    const Pt shift = cp_add(cp_neg(coset.initial), point_of_index(coset.step_size >> 1));
    Var x = pq_add_const(p, shift).x;
    for (uint32_t i = 1; i < coset.log_size; i++) {
        const Var sq = mul(x, x);
        const Var dbl = add(sq, sq);
        x = sub(dbl, m31_one(cs));
    }
    return x;
}
inline Var combine_ef(const Var& v0, const Var& v1, const Var& v2, const Var& v3) {
    const Var s1 = add(v0, shift_by_i(v1));
    const Var s2 = add(s1, shift_by_j(v2));
    return add(s2, shift_by_ij(v3));
}
struct Accumulator {
    Var random_coeff, accumulation;
    void accumulate(const Var& evaluation) { accumulation = add(mul(accumulation, random_coeff), evaluation); }
};
// EvalAtRowVar + LogupAtRowVar (composition/src/data_structures.rs:57-215)
struct EvalAtRow {
    size_t col_index[3] = {0, 0, 0};
    const std::vector<std::vector<Var>>* mask[3];
    size_t base[3];  // first column of this component inside each tree
    Var cumsum_shift, denom_inverse;
    std::vector<std::pair<Var, Var>> fracs;
    Accumulator* acc;
    const std::vector<Var>& next_mask(int interaction) { return (*mask[interaction])[base[interaction] + col_index[interaction]++]; }
    Var next_trace_mask() { return next_mask(1)[0]; }
    Var get_preprocessed_column() { return next_mask(0)[0]; }
    std::vector<Var> next_extension_interaction_mask(size_t n) {
        const std::vector<Var>* cols[4];
        for (int k = 0; k < 4; k++) cols[k] = &next_mask(2);
        std::vector<Var> out;
        for (size_t s = 0; s < n; s++) out.push_back(combine_ef((*cols[0])[s], (*cols[1])[s], (*cols[2])[s], (*cols[3])[s]));
        return out;
    }
    void add_to_relation(const Lookup& lk, const Var& multiplicity, const std::vector<Var>& values) {
        Var denom = mul(lk.alpha_powers[0], values[0]);
        for (size_t k = 1; k < values.size(); k++) denom = add(denom, mul(lk.alpha_powers[k], values[k]));
        denom = sub(denom, lk.z);
        fracs.push_back({multiplicity, denom});
    }
    void add_constraint(const Var& value) { acc->accumulate(mul(value, denom_inverse)); }
    void finalize_logup(size_t batch_size) {
        ConstraintSystem* cs = cumsum_shift.cs;
        std::vector<std::pair<Var, Var>> batched;
        for (size_t k = 0; k < fracs.size(); k += batch_size) {
            const size_t end = std::min(fracs.size(), k + batch_size);
            if (end - k == 1) { batched.push_back(fracs[k]); continue; }
            Var p = fracs[k].first, q = fracs[k].second;
            for (size_t e = k + 1; e < end; e++) {
                const Var pq = mul(p, fracs[e].second);
                const Var mq = mul(fracs[e].first, q);
                p = add(pq, mq);
                q = mul(q, fracs[e].second);
            }
            batched.push_back({p, q});
        }
        Var prev_col_cumsum = qm31_zero(cs);
        for (size_t k = 0; k + 1 < batched.size(); k++) {
            const Var cur = next_extension_interaction_mask(1)[0];
            const Var diff = sub(cur, prev_col_cumsum);
            prev_col_cumsum = cur;
            add_constraint(sub(mul(diff, batched[k].second), batched[k].first));
        }
        {
            const auto pc = next_extension_interaction_mask(2);
            const Var d1 = sub(pc[1], pc[0]);
            const Var diff = sub(d1, prev_col_cumsum);
            const Var fixed = add(diff, cumsum_shift);
            add_constraint(sub(mul(fixed, batched.back().second), batched.back().first));
        }
    }
};
inline EvalAtRow make_eval(const ProofVar& pv, bool poseidon, const Var& total_sum, const Var& denom_inverse, uint32_t log_size, Accumulator* acc) {
    EvalAtRow ev;
    for (int t = 0; t < 3; t++) {
        ev.mask[t] = &pv.sampled_values[t];
        ev.base[t] = poseidon ? PLONK_COLS_[t] : 0;
    }
    ev.cumsum_shift = mul_constant(total_sum, h_inv(h_pow(2, log_size)));
    ev.denom_inverse = denom_inverse;
    ev.acc = acc;
    return ev;
}

inline void evaluate_plonk(const Lookup& lk, EvalAtRow& ev) {  // composition/src/plonk.rs:8-82
    ConstraintSystem* cs = lk.z.cs;
    const Var a_wire = ev.get_preprocessed_column(), b_wire = ev.get_preprocessed_column(), c_wire = ev.get_preprocessed_column();
    const Var op = ev.get_preprocessed_column();
    const Var mult_a = ev.get_preprocessed_column(), mult_b = ev.get_preprocessed_column(), mult_c = ev.get_preprocessed_column();
    const Var poseidon_wire = ev.get_preprocessed_column(), mult_poseidon = ev.get_preprocessed_column(), enforce_c_m31 = ev.get_preprocessed_column();
    Var a_v[4], b_v[4], c_v[4];
    for (auto& v : a_v) v = ev.next_trace_mask();
    for (auto& v : b_v) v = ev.next_trace_mask();
    for (auto& v : c_v) v = ev.next_trace_mask();
    ev.add_constraint(mul(enforce_c_m31, c_v[1]));
    ev.add_constraint(mul(enforce_c_m31, c_v[2]));
    ev.add_constraint(mul(enforce_c_m31, c_v[3]));
    const Var a_val = combine_ef(a_v[0], a_v[1], a_v[2], a_v[3]);
    const Var b_val = combine_ef(b_v[0], b_v[1], b_v[2], b_v[3]);
    const Var c_val = combine_ef(c_v[0], c_v[1], c_v[2], c_v[3]);
    const Var ab = add(a_val, b_val);
    const Var t1 = sub(c_val, mul(op, ab));
    const Var one_minus_op = sub(qm31_one(cs), op);
    const Var t2 = mul(mul(one_minus_op, a_val), b_val);
    ev.add_constraint(sub(t1, t2));
    ev.add_to_relation(lk, mult_a, {a_val, a_wire});
    ev.add_to_relation(lk, mult_b, {b_val, b_wire});
    ev.add_to_relation(lk, mult_c, {c_val, c_wire});
    const Var neg_mp = neg(mult_poseidon);
    ev.add_to_relation(lk, neg_mp, {poseidon_wire, a_val, b_val});
    ev.finalize_logup(2);
}

inline void apply_m4(Var* x) {
    const Var t0 = add(x[0], x[1]);
    const Var t02 = add(t0, t0);
    const Var t1 = add(x[2], x[3]);
    const Var t12 = add(t1, t1);
    const Var t2 = add(add(x[1], x[1]), t1);
    const Var t3 = add(add(x[3], x[3]), t0);
    const Var t4 = add(add(t12, t12), t3);
    const Var t5 = add(add(t02, t02), t2);
    const Var t6 = add(t3, t5);
    const Var t7 = add(t2, t4);
    x[0] = t6; x[1] = t5; x[2] = t7; x[3] = t4;
}
inline void apply_external_round_matrix(Var* state) {
    for (int i = 0; i < 4; i++) apply_m4(state + 4 * i);
    for (int j = 0; j < 4; j++) {
        const Var s = add(add(add(state[j], state[j + 4]), state[j + 8]), state[j + 12]);
        for (int i = 0; i < 4; i++) state[4 * i + j] = add(state[4 * i + j], s);
    }
}
inline void apply_internal_round_matrix(Var* state) {
    Var total = state[0];
    for (int i = 1; i < 16; i++) total = add(total, state[i]);
    const Var dbl = add(state[0], state[0]);
    state[0] = add(state[0], add(dbl, total));
    for (int i = 1; i < 16; i++) state[i] = add(mul_constant(state[i], 1u << (i + 1)), total);
}
inline Var pow5(const Var& x) {
    const Var x2 = mul(x, x);
    const Var x4 = mul(x2, x2);
    return mul(x4, x);
}

inline void evaluate_poseidon(const Lookup& lk, EvalAtRow& ev) {  // composition/src/poseidon.rs:73-241
    ConstraintSystem* cs = lk.z.cs;
    const Var is_first_round = ev.get_preprocessed_column(), is_last_round = ev.get_preprocessed_column(), is_full_round = ev.get_preprocessed_column();
    const Var one = qm31_one(cs);
    const Var is_not_first_round = sub(one, is_first_round);
    const Var is_not_last_round = sub(one, is_last_round);
    const Var is_partial_round = sub(is_not_first_round, is_full_round);
    const Var round_id = ev.get_preprocessed_column();
    Var rc0[16], rc1[16];
    for (auto& v : rc0) v = ev.get_preprocessed_column();
    for (auto& v : rc1) v = ev.get_preprocessed_column();
    const Var external_idx_1 = ev.get_preprocessed_column(), external_idx_2 = ev.get_preprocessed_column();
    const Var is_external_idx_1_nonzero = ev.get_preprocessed_column(), is_external_idx_2_nonzero = ev.get_preprocessed_column();
    const Var swap_bit_addr = rc0[0];
    Var in_state[16], intermediate_state[16], out_state[16];
    for (auto& v : in_state) v = ev.next_trace_mask();
    for (auto& v : intermediate_state) v = ev.next_trace_mask();
    for (auto& v : out_state) v = ev.next_trace_mask();
    const Var swap_bit_value = intermediate_state[0];

    const Var one_minus_swap = sub(one, swap_bit_value);
    Var permuted[16];
    for (int i = 0; i < 16; i++) {
        if (i < 8) {
            const Var l = mul(in_state[i], one_minus_swap);
            const Var r = mul(in_state[i + 8], swap_bit_value);
            permuted[i] = add(l, r);
        } else {
            const Var l = mul(in_state[i - 8], swap_bit_value);
            const Var r = mul(in_state[i], one_minus_swap);
            permuted[i] = add(l, r);
        }
    }
    apply_external_round_matrix(permuted);
    for (int i = 0; i < 16; i++) ev.add_constraint(mul(is_first_round, sub(permuted[i], out_state[i])));

    Var full[16];
    for (int i = 0; i < 16; i++) full[i] = add(in_state[i], rc0[i]);
    for (int i = 0; i < 16; i++) full[i] = pow5(full[i]);
    for (int i = 0; i < 16; i++) {
        ev.add_constraint(mul(is_full_round, sub(intermediate_state[i], full[i])));
        full[i] = intermediate_state[i];
    }
    apply_external_round_matrix(full);
    for (int i = 0; i < 16; i++) full[i] = add(full[i], rc1[i]);
    for (int i = 0; i < 16; i++) full[i] = pow5(full[i]);
    apply_external_round_matrix(full);
    for (int i = 0; i < 16; i++) ev.add_constraint(mul(is_full_round, sub(out_state[i], full[i])));

    Var partial[16];
    for (int i = 0; i < 16; i++) partial[i] = in_state[i];
    for (int r = 0; r < 14; r++) {
        partial[0] = add(partial[0], rc0[r]);
        partial[0] = pow5(partial[0]);
        ev.add_constraint(mul(is_partial_round, sub(intermediate_state[r], partial[0])));
        partial[0] = intermediate_state[r];
        apply_internal_round_matrix(partial);
    }
    for (int i = 0; i < 16; i++) ev.add_constraint(mul(is_partial_round, sub(out_state[i], partial[i])));

    const Var in_left_id = add(round_id, round_id);
    const Var in_right_id = add(in_left_id, one);
    const Var out_left_id = add(in_right_id, one);
    const Var out_right_id = add(out_left_id, one);

    auto relation = [&](const Var& nonzero, const Var& is_round, const Var& is_not_round, const Var& ext_idx, const Var& inner_id, const Var* st,
                        bool plus) {
        const Var sel = mul(nonzero, is_round);
        const Var i1 = mul(is_round, ext_idx);
        const Var i2 = mul(is_not_round, inner_id);
        const Var ident = add(i1, i2);
        const Var a = combine_ef(st[0], st[1], st[2], st[3]);
        const Var b = combine_ef(st[4], st[5], st[6], st[7]);
        const Var multiplicity = plus ? add(sel, is_not_round) : sub(sel, is_not_round);
        ev.add_to_relation(lk, multiplicity, {ident, a, b});
    };
    relation(is_external_idx_1_nonzero, is_first_round, is_not_first_round, external_idx_1, in_left_id, in_state, false);
    relation(is_external_idx_2_nonzero, is_first_round, is_not_first_round, external_idx_2, in_right_id, in_state + 8, false);
    relation(is_external_idx_1_nonzero, is_last_round, is_not_last_round, external_idx_1, out_left_id, out_state, true);
    relation(is_external_idx_2_nonzero, is_last_round, is_not_last_round, external_idx_2, out_right_id, out_state + 8, true);
    const Var sw_mult = mul(is_first_round, is_not_last_round);
    ev.add_to_relation(lk, sw_mult, {swap_bit_value, swap_bit_addr});
    ev.finalize_logup(3);
}

inline void composition_check(const ProofVar& pv, const Template& d, const FiatShamir& fs) {  // composition/src/lib.rs:33-129
    Accumulator acc{fs.random_coeff, qm31_zero(fs.random_coeff.cs)};
    {
        const Var vanish_inv = qm31_inv(coset_vanishing(fs.oods_point, d.lp));
        EvalAtRow ev = make_eval(pv, false, pv.plonk_total_sum, vanish_inv, d.lp, &acc);
        evaluate_plonk(fs.lookup, ev);
    }
    {
        const Var vanish_inv = qm31_inv(coset_vanishing(fs.oods_point, d.lq));
        EvalAtRow ev = make_eval(pv, true, pv.poseidon_total_sum, vanish_inv, d.lq, &acc);
        evaluate_poseidon(fs.lookup, ev);
    }
    const auto& sv = pv.sampled_values[3];
    const Var left = combine_ef(sv[0][0], sv[1][0], sv[2][0], sv[3][0]);
    const Var right = combine_ef(sv[4][0], sv[5][0], sv[6][0], sv[7][0]);
    const uint32_t comp_log_degree_bound = std::max(d.lp + 1, d.lq + 2) + 1;
    const Var dbl = pq_repeated_double_x_only(fs.oods_point, comp_log_degree_bound - 2);
    const Var expected = add(left, mul(right, dbl));
    equalverify(acc.accumulation, expected);
}

// ---------------------------------------------------------------- SinglePath / SinglePair Merkle proofs
struct PathProofVar {  // SinglePathMerkleProofVar (data_structures/src/lib.rs:283-354)
    uint32_t query, depth;
    std::vector<Half> sibling_hashes;
    std::map<uint32_t, std::vector<Var>> columns;
};
inline PathProofVar path_proof_new(ConstraintSystem* cs, const Template& d, int t, uint32_t i) {
    PathProofVar p;
    p.depth = t == 3 ? d.M : std::max(d.A, d.B);
    p.query = d.trace_pos[(size_t)t * d.nq + i];
    for (uint32_t k = 0; k < p.depth; k++) p.sibling_hashes.push_back(half_single_use(cs, words8(d.trace_sib + (((size_t)t * d.nq + i) * d.M + k) * 8)));
    // ascending log size (BTreeMap order); a level's values sit behind those of the levels above it in the 64-word row
    auto levels = d.tree_levels(t);
    std::map<uint32_t, std::pair<uint32_t, uint32_t>> where;  // log size -> (offset, count)
    uint32_t off = 0;
    for (auto [ls, n] : levels) { where[ls] = {off, n}; off += n; }
    for (auto& [ls, oc] : where) {
        std::vector<Var> vals;
        for (uint32_t j = 0; j < oc.second; j++)
            vals.push_back(m31_witness(cs, d.trace_cols[((size_t)t * d.nq + i) * 64 + oc.first + j], mk_instr(W_TRACE_COL, 0, 0, (uint32_t)t, i, oc.first + j)));
        p.columns[ls] = vals;
    }
    return p;
}
inline void path_proof_verify(Gadgets& g, PathProofVar& p, const Half& root, const Bits& query) {
    if (query.get_value() != p.query % MP) throw std::runtime_error("query position does not match the path hint");
    Half cur = g.hash_m31_columns_get_rate(p.columns.at(p.depth));
    for (uint32_t i = 0; i < p.depth; i++) {
        const uint32_t h = p.depth - i - 1;
        auto it = p.columns.find(h);
        if (it != p.columns.end()) {
            const Half column_hash = g.hash_m31_columns_get_capacity(it->second);
            cur = g.hash_tree_with_column_hash_with_swap(cur, p.sibling_hashes[i], query.value[i], query.variables[i], column_hash);
        } else {
            cur = g.hash_tree_with_swap(cur, p.sibling_hashes[i], query.value[i], query.variables[i]);
        }
    }
    if (cur.value != root.value) throw std::runtime_error("Merkle path does not reach the root");
    g.half_equalverify(cur, root);
}

struct PairProofVar {  // SinglePairMerkleProofVar (data_structures/src/lib.rs:357-464)
    uint32_t query, depth;
    std::vector<Half> sibling_hashes;
    std::map<uint32_t, Var> self_columns, siblings_columns;
};
inline PairProofVar pair_proof_new(ConstraintSystem* cs, const Template& d, uint32_t tree, uint32_t i) {
    PairProofVar p;
    p.depth = tree == 0 ? d.M : d.M - tree;
    p.query = d.trace_pos[(size_t)3 * d.nq + i] >> (d.M - p.depth);
    for (uint32_t k = 0; k + 1 < p.depth; k++) p.sibling_hashes.push_back(half_single_use(cs, words8(d.fri_sib + (((size_t)tree * d.nq + i) * d.M + k) * 8)));
    std::vector<uint32_t> levels;  // data levels, descending (the c-th from the top is pair c of d_fri_cols)
    if (tree == 0) {
        auto all = d.all_log_sizes();
        levels.assign(all.rbegin(), all.rend());
    } else levels = {p.depth};
    std::map<uint32_t, uint32_t> where;
    for (uint32_t c = 0; c < levels.size(); c++) where[levels[c]] = c;
    const uint32_t* cols = d.fri_cols + ((size_t)tree * d.nq + i) * 24;
    for (auto [ls, c] : where) p.self_columns.emplace(ls, qm31_witness(cs, words4(cols + 8 * c), mk_instr(W_FRI_COL, 0, 0, tree, i, 8 * c)));
    for (auto [ls, c] : where) p.siblings_columns.emplace(ls, qm31_witness(cs, words4(cols + 8 * c + 4), mk_instr(W_FRI_COL, 0, 0, tree, i, 8 * c + 4)));
    return p;
}
inline void pair_proof_verify(Gadgets& g, PairProofVar& p, const Half& root, const Bits& query) {
    ConstraintSystem* cs = g.cs;
    if (query.get_value() != p.query % MP) throw std::runtime_error("query position does not match the pair hint");
    Half self_hash = g.hash_qm31_pair_get_rate(p.self_columns.at(p.depth), qm31_zero(cs));
    Half sibling_hash = g.hash_qm31_pair_get_rate(p.siblings_columns.at(p.depth), qm31_zero(cs));
    for (uint32_t i = 0; i < p.depth; i++) {
        const uint32_t h = p.depth - i - 1;
        auto it = p.self_columns.find(h);
        if (it == p.self_columns.end()) {
            self_hash = g.hash_tree_with_swap(self_hash, sibling_hash, query.value[i], query.variables[i]);
            if (i != p.depth - 1) sibling_hash = p.sibling_hashes[i];
        } else {
            const Half self_column_hash = g.hash_qm31_pair_get_capacity(it->second, qm31_zero(cs));
            const Half sibling_column_hash = g.hash_qm31_pair_get_capacity(p.siblings_columns.at(h), qm31_zero(cs));
            self_hash = g.hash_tree_with_column_hash_with_swap(self_hash, sibling_hash, query.value[i], query.variables[i], self_column_hash);
            sibling_hash = g.combine_hash_tree_with_column(p.sibling_hashes[i], sibling_column_hash);
        }
    }
    if (self_hash.value != root.value) throw std::runtime_error("Merkle pair path does not reach the root");
    g.half_equalverify(self_hash, root);
}

// ---------------------------------------------------------------- AnswerResults::compute
// key 0 = ShiftIndex::Zero, 1 = Shift(-1, lp), 2 = Shift(-1, lq) — ONE key when the two components have the same log size
// (the reference's key is the pair (shift, log size), answer/src/data_structures.rs:14-27)
struct Sample { int key; const PointQM31* point; Var value; };
struct Answer {
    std::map<uint32_t, std::vector<PointCarryingQuery>> qp;
    std::vector<uint32_t> all_log_sizes;  // ascending
    std::vector<std::vector<Var>> fri_answers;  // per log size, descending
};

inline std::array<Var, 3> complex_conjugate_line_coeffs(const PointQM31& point, const Var& value, const Var& alpha) {
    const auto v = decompose_cm31(value);
    const auto y = decompose_cm31(point.y);
    const Var v0y1 = mul(v[0], y[1]);
    const Var v1y0 = mul(v[1], y[0]);
    const Var b = sub(v0y1, v1y0);
    const Var ra = mul(alpha, v[1]);
    const Var rb = mul(alpha, b);
    const Var rc = mul(alpha, y[1]);
    return {ra, rb, rc};
}

// answer/src/lib.rs:366-396 with answer/src/data_structures.rs:43-215
inline std::vector<Var> fri_answers_for_log_size(ConstraintSystem* cs, const std::vector<std::vector<Sample>>& samples, const Var& random_coeff,
                                                 const std::vector<PointCarryingQuery>& query_positions, const std::vector<std::vector<Var>>& queried_values) {
    struct Batch { const PointQM31* point; std::vector<std::pair<size_t, Var>> cvs; };
    std::vector<int> order;  // IndexMap: insertion order of the keys
    std::map<int, Batch> batches;
    for (size_t column = 0; column < samples.size(); column++)
        for (const Sample& s : samples[column]) {
            if (!batches.count(s.key)) { order.push_back(s.key); batches[s.key].point = s.point; }
            batches[s.key].cvs.push_back({column, s.value});
        }
    Var alpha = qm31_constant(cs, {0, 0, MP - 2, 0});
    std::vector<std::vector<std::array<Var, 3>>> line_coeffs;
    for (int key : order) {
        const Batch& b = batches[key];
        std::vector<std::array<Var, 3>> per;
        for (const auto& cv : b.cvs) {
            per.push_back(complex_conjugate_line_coeffs(*b.point, cv.second, alpha));
            alpha = mul(alpha, random_coeff);
        }
        line_coeffs.push_back(per);
    }
    std::vector<Var> evals;
    for (size_t qi = 0; qi < query_positions.size(); qi++) {
        const PointM31 domain_point = query_positions[qi].get_next_point();
        const std::vector<Var>& row = queried_values[qi];
        std::vector<Var> denominator_inverses;
        for (int key : order) {
            const PointQM31& point = *batches[key].point;
            const auto px = decompose_cm31(point.x);
            const auto py = decompose_cm31(point.y);
            const Var a = mul(sub(px[0], domain_point.x), py[1]);
            const Var b = mul(sub(py[0], domain_point.y), px[1]);
            denominator_inverses.push_back(cm31_inv(sub(a, b)));
        }
        Var row_acc = qm31_zero(cs);
        for (size_t bi = 0; bi < order.size(); bi++) {
            const Batch& b = batches[order[bi]];
            Var numerator = qm31_zero(cs);
            for (size_t k = 0; k < b.cvs.size(); k++) {
                const auto& co = line_coeffs[bi][k];
                const Var value = mul(row[b.cvs[k].first], co[2]);
                const Var linear_term = add(mul(co[0], domain_point.y), co[1]);
                numerator = add(numerator, sub(value, linear_term));
            }
            row_acc = add(row_acc, mul(numerator, denominator_inverses[bi]));
        }
        evals.push_back(row_acc);
    }
    return evals;
}

inline Answer answer(Gadgets& g, const ProofVar& pv, const Template& d, const FiatShamir& fs, uint32_t walk) {
    ConstraintSystem* cs = g.cs;
    Answer ans;
    // CirclePointQM31Var::new_witness(&cs, &fiat_shamir_hints.oods_point) (examples/multi-proofs/src/main.rs:108)
    PointQM31 oods_point;
    oods_point.x = qm31_witness(cs, fs.oods_point.x.value, mk_instr(W_COPY, fs.oods_point.x.variable));
    oods_point.y = qm31_witness(cs, fs.oods_point.y.value, mk_instr(W_COPY, fs.oods_point.y.variable));
    const Pt step_plonk = canonic_coset(d.lp).step, step_poseidon = canonic_coset(d.lq).step;
    PointQM31 shifted_plonk[2], shifted_poseidon[2];  // [0] = shift 0, [1] = shift -1
    for (int k = 0; k < 2; k++) {
        const int which = (walk & 1u) ? 1 - k : k;
        shifted_plonk[which] = pq_add_const(oods_point, cp_mul(step_plonk, which ? 0x7fffffffu : 0u));
    }
    for (int k = 0; k < 2; k++) {
        const int which = (walk & 2u) ? 1 - k : k;
        shifted_poseidon[which] = pq_add_const(oods_point, cp_mul(step_poseidon, which ? 0x7fffffffu : 0u));
    }

    // samples[tree][column] = [(shift key, point, value)]
    std::vector<std::vector<std::vector<Sample>>> samples(4);
    for (int t = 0; t < 3; t++) {
        for (int comp = 0; comp < 2; comp++) {
            const uint32_t n_cols = comp ? POSEIDON_COLS_[t] : PLONK_COLS_[t], base = comp ? PLONK_COLS_[t] : 0;
            const PointQM31* shifted = comp ? shifted_poseidon : shifted_plonk;
            for (uint32_t c = 0; c < n_cols; c++) {
                const auto& vals = pv.sampled_values[t][base + c];
                std::vector<Sample> col;
                if (t == 0) col.push_back({0, &oods_point, vals[0]});
                else if (t == 2 && c >= 4) { col.push_back({(comp && d.lp != d.lq) ? 2 : 1, &shifted[1], vals[0]}); col.push_back({0, &shifted[0], vals[1]}); }
                else col.push_back({0, &shifted[0], vals[0]});
                samples[t].push_back(col);
            }
        }
    }
    for (uint32_t c = 0; c < 8; c++) samples[3].push_back({Sample{0, &oods_point, pv.sampled_values[3][c][0]}});

    const uint32_t A = d.A, B = d.B, M = d.M, min_degree = d.log_last + d.blowup + 1;
    // QueryPositionsPerLogSizeVar::new (query/src/lib.rs:19-48)
    std::vector<PointCarryingQuery> elems;
    for (const Var& q : fs.raw_queries) elems.push_back(point_carrying_query(bits_from_m31(q, 31).range(0, M)));
    ans.qp[M] = elems;
    for (uint32_t log_size = M; log_size-- > min_degree;) {
        for (auto& e : elems) e.next();
        ans.qp[log_size] = elems;
    }
    ans.all_log_sizes = d.all_log_sizes();
    // DecommitmentVar::new, then the four trees' paths (answer/src/lib.rs:212-262)
    std::vector<std::vector<PathProofVar>> dec(4);
    for (int t = 0; t < 4; t++)
        for (uint32_t i = 0; i < d.nq; i++) dec[t].push_back(path_proof_new(cs, d, t, i));
    for (int t = 0; t < 3; t++)
        for (uint32_t i = 0; i < d.nq; i++) path_proof_verify(g, dec[t][i], pv.commitments[t], ans.qp[std::max(A, B)][i].bits);
    for (uint32_t i = 0; i < d.nq; i++) path_proof_verify(g, dec[3][i], pv.commitments[3], ans.qp[M][i].bits);
    const uint32_t col_size_of[3][2] = {{A, B}, {A, B}, {A, B}};
    for (auto it = ans.all_log_sizes.rbegin(); it != ans.all_log_sizes.rend(); ++it) {
        const uint32_t ls = *it;
        std::vector<std::vector<Var>> queried_values(d.nq);
        for (uint32_t i = 0; i < d.nq; i++)
            for (int t = 0; t < 4; t++) {
                auto f = dec[t][i].columns.find(ls);
                if (f != dec[t][i].columns.end()) queried_values[i].insert(queried_values[i].end(), f->second.begin(), f->second.end());
            }
        std::vector<std::vector<Sample>> group;
        for (int t = 0; t < 3; t++)
            for (size_t c = 0; c < samples[t].size(); c++)
                if (col_size_of[t][c >= PLONK_COLS_[t]] == ls) group.push_back(samples[t][c]);
        if (ls == M)
            for (const auto& col : samples[3]) group.push_back(col);
        ans.fri_answers.push_back(fri_answers_for_log_size(cs, group, fs.after_sampled_values_random_coeff, ans.qp[ls], queried_values));
    }
    return ans;
}

// ---------------------------------------------------------------- FoldingResults::compute (folding/src/lib.rs:11-206)
inline void folding(Gadgets& g, const ProofVar& pv, const Template& d, const FiatShamir& fs, Answer& ans) {
    ConstraintSystem* cs = g.cs;
    const uint32_t M = d.M;
    std::vector<PairProofVar> proofs;
    for (uint32_t i = 0; i < d.nq; i++) {
        proofs.push_back(pair_proof_new(cs, d, 0, i));
        pair_proof_verify(g, proofs.back(), pv.first_layer_commitment, ans.qp[M][i].bits);
    }
    {
        size_t k = 0;
        for (auto it = ans.all_log_sizes.rbegin(); it != ans.all_log_sizes.rend(); ++it, ++k)
            for (uint32_t i = 0; i < d.nq; i++) equalverify(proofs[i].self_columns.at(*it), ans.fri_answers[k][i]);
    }
    std::map<uint32_t, std::vector<Var>> folded_results;
    for (uint32_t ls : ans.all_log_sizes) {
        std::vector<Var> per;
        for (uint32_t i = 0; i < d.nq; i++) {
            const PointCarryingQuery& query = ans.qp[ls][i];
            const Var& self_val = proofs[i].self_columns.at(ls);
            const Var& sibling_val = proofs[i].siblings_columns.at(ls);
            const PointM31 point = pm_double(query.point);
            const Var y_inv = m31_inv(point.y);
            const auto lr = swap(self_val, sibling_val, query.bits.value[0], query.bits.variables[0]);
            const Var new_left = add(lr.first, lr.second);
            const Var new_right = mul(sub(lr.first, lr.second), y_inv);
            per.push_back(add(new_left, mul(new_right, fs.fri_alphas[M - ls])));
        }
        folded_results[ls] = per;
    }
    uint32_t log_size = M;
    std::vector<Var> folded(d.nq, qm31_zero(cs));
    for (uint32_t i = 0; i < d.n_inner; i++) {
        auto f = folded_results.find(log_size);
        if (f != folded_results.end()) {
            const Var fri_alpha = mul(fs.fri_alphas[i], fs.fri_alphas[i]);
            for (uint32_t k = 0; k < d.nq; k++) folded[k] = add(mul(fri_alpha, folded[k]), f->second[k]);
        }
        log_size--;
        std::vector<Var> new_folded;
        for (uint32_t k = 0; k < d.nq; k++) {
            const PointCarryingQuery& query = ans.qp[log_size][k];
            PairProofVar merkle_proof = pair_proof_new(cs, d, 1 + i, k);
            const Var& self_val = merkle_proof.self_columns.at(log_size);
            const Var& sibling_val = merkle_proof.siblings_columns.at(log_size);
            equalverify(folded[k], self_val);
            const Var x_inv = m31_inv(query.point.x);
            const auto lr = swap(self_val, sibling_val, query.bits.value[0], query.bits.variables[0]);
            const Var new_left = add(lr.first, lr.second);
            const Var new_right = mul(sub(lr.first, lr.second), x_inv);
            new_folded.push_back(add(new_left, mul(new_right, fs.fri_alphas[i + 1])));
            pair_proof_verify(g, merkle_proof, pv.inner_layer_commitments[i], query.bits);
        }
        folded = new_folded;
    }
    for (uint32_t k = 0; k < d.nq; k++) {
        if (pv.last_poly.size() == 1) equalverify(folded[k], pv.last_poly[0]);
        else {
            const Var x = ans.qp[log_size][k].get_next_point_x();
            equalverify(folded[k], line_eval_at_point(cs, pv.last_poly, x));
        }
    }
}

// One copy of the verifier: the body of the `multipliers` loop (examples/multi-proofs/src/main.rs:66-139)
inline void verify_in_circuit(Gadgets& g, const Template& d, const std::vector<std::pair<uint32_t, Q4>>& public_inputs, uint32_t walk) {
    ConstraintSystem* cs = g.cs;
    std::vector<std::pair<uint32_t, Var>> inputs;
    for (const auto& [idx, val] : public_inputs) inputs.push_back({idx, qm31_constant(cs, val)});
    ProofVar pv = allocate_proof(cs, d);
    const FiatShamir fs = fiat_shamir(g, pv, d, inputs);
    composition_check(pv, d, fs);
    Answer ans = answer(g, pv, d, fs, walk);
    folding(g, pv, d, fs, ans);
}

}  // namespace rsv::circuit
