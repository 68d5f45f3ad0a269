// circuit_cs.hpp — HOST code: the reference's Plonk-with-Poseidon constraint system and its field variables, mirrored so
// that a run of the circuit's gadgets can be written down as a witness program (rsv_witness_program_build, include/rsv.h).
//
// What the reference's recursion circuit leaves behind for the next prover is `variables: Vec<QM31>` plus the gate lists
// and the PoseidonFlow (constraint_system/src/plonk_with_poseidon.rs:17-41).  Every gadget call appends to them in program
// order (:140-283), so the vector is only reproduced by replaying the gadgets in that order.  This is done ONCE per proof
// shape, on a template proof, with host integers; kept for every variable is HOW it came to be: its instruction
// (k_witness.hpp WitnessOp).  M31Var / CM31Var / QM31Var: primitives/fields/src/{m31,cm31,qm31}.rs — operand order as
// there, because it decides the wire order.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace rsv::circuit {

constexpr uint32_t MP = 0x7fffffffu;
using Q4 = std::array<uint32_t, 4>;  // QM31 (a0, a1, a2, a3) = (a0 + a1 i) + (a2 + a3 i) j

inline uint32_t h_add(uint32_t a, uint32_t b) { uint32_t s = a + b; return s >= MP ? s - MP : s; }
inline uint32_t h_sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + MP - b; }
inline uint32_t h_neg(uint32_t a) { return a ? MP - a : 0; }
inline uint32_t h_mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % MP); }
inline uint32_t h_pow(uint32_t a, uint32_t e) {
    uint32_t r = 1;
    while (e) { if (e & 1) r = h_mul(r, a); a = h_mul(a, a); e >>= 1; }
    return r;
}
inline uint32_t h_inv(uint32_t a) { return h_pow(a, MP - 2); }
struct C2 { uint32_t re, im; };
inline C2 hc_mul(C2 a, C2 b) { return {h_sub(h_mul(a.re, b.re), h_mul(a.im, b.im)), h_add(h_mul(a.re, b.im), h_mul(a.im, b.re))}; }
inline C2 hc_inv(C2 a) {
    const uint32_t d = h_inv(h_add(h_mul(a.re, a.re), h_mul(a.im, a.im)));
    return {h_mul(a.re, d), h_mul(h_neg(a.im), d)};
}
inline Q4 hq_add(const Q4& a, const Q4& b) { return {h_add(a[0], b[0]), h_add(a[1], b[1]), h_add(a[2], b[2]), h_add(a[3], b[3])}; }
inline Q4 hq_neg(const Q4& a) { return {h_neg(a[0]), h_neg(a[1]), h_neg(a[2]), h_neg(a[3])}; }
inline Q4 hq_scale(const Q4& a, uint32_t k) { return {h_mul(a[0], k), h_mul(a[1], k), h_mul(a[2], k), h_mul(a[3], k)}; }
inline Q4 hq_mul(const Q4& a, const Q4& b) {  // (A + B j)(C + D j) = AC + (2 + i) BD + (AD + BC) j
    const C2 A{a[0], a[1]}, B{a[2], a[3]}, Cc{b[0], b[1]}, D{b[2], b[3]};
    const C2 ac = hc_mul(A, Cc), bd = hc_mul(B, D), ad = hc_mul(A, D), bc = hc_mul(B, Cc);
    return {h_sub(h_add(ac.re, h_add(bd.re, bd.re)), bd.im), h_add(h_add(ac.im, h_add(bd.im, bd.im)), bd.re), h_add(ad.re, bc.re),
            h_add(ad.im, bc.im)};
}
inline Q4 hq_inv(const Q4& a) {  // 1 / (A + B j) = (A - B j) / (A^2 - (2 + i) B^2)
    const C2 A{a[0], a[1]}, B{a[2], a[3]};
    const C2 b2 = hc_mul(B, B), a2 = hc_mul(A, A);
    const C2 ib2{h_sub(h_add(b2.re, b2.re), b2.im), h_add(h_add(b2.im, b2.im), b2.re)};
    const C2 den = hc_inv({h_sub(a2.re, ib2.re), h_sub(a2.im, ib2.im)});
    const C2 r0 = hc_mul(A, den), r1 = hc_mul({h_neg(B.re), h_neg(B.im)}, den);
    return {r0.re, r0.im, r1.re, r1.im};
}

struct Instr { uint32_t op, dst, a, b, imm[4]; };  // = 8 words of a witness program
inline Instr mk_instr(uint32_t op, uint32_t a = 0, uint32_t b = 0, uint32_t i0 = 0, uint32_t i1 = 0, uint32_t i2 = 0, uint32_t i3 = 0) {
    return Instr{op, 0, a, b, {i0, i1, i2, i3}};
}

struct FlowRecord {           // one Poseidon2HalfVar::permute invocation (plonk_with_poseidon.rs:117-128)
    uint32_t wire[4];         // PoseidonEntry::wire of r1..r4
    uint32_t addr;            // SwapOption::addr
};

enum Mode { WITNESS, CONSTANT };

struct GateRow { uint32_t a, b, c, op, poseidon_wire, enforce_c_m31; };  // one Plonk row (plonk_with_poseidon.rs:23-36)
struct WitnessOp3 { uint32_t row, bit, constant; };  // a row whose `op` follows the witness: op = variables[bit] ? constant : 0

// PlonkWithPoseidonConstraintSystem (plonk_with_poseidon.rs:17-283): the values, how each came to be, the gate rows, the
// flow's wires.
struct ConstraintSystem {
    std::vector<Q4> variables;
    std::vector<Instr> origin;
    std::vector<FlowRecord> flow;
    std::vector<GateRow> rows;
    std::vector<WitnessOp3> witness_ops;
    std::unordered_map<std::string, uint32_t> cache;
    bool have_hint = false;
    Instr hint{};

    ConstraintSystem() {
        variables = {Q4{0, 0, 0, 0}, Q4{1, 0, 0, 0}, Q4{0, 1, 0, 0}, Q4{0, 0, 1, 0}};
        for (uint32_t k = 0; k < 4; k++) {
            Instr in = mk_instr(W_CONST, 0, 0, variables[k][0], variables[k][1], variables[k][2], variables[k][3]);
            in.dst = k;
            origin.push_back(in);
            rows.push_back(GateRow{k, 0, k, 1, 0, 0});
        }
    }
    void row(uint32_t a, uint32_t b, uint32_t c, uint32_t op, uint32_t pw = 0, uint32_t m31 = 0) { rows.push_back(GateRow{a, b, c, op, pw, m31}); }
    uint32_t push(const Q4& v, Instr in) {
        in.dst = (uint32_t)variables.size();
        variables.push_back(v);
        origin.push_back(in);
        return in.dst;
    }
    void insert_gate(uint32_t a, uint32_t b, uint32_t c, uint32_t op) { row(a, b, c, op); }
    void enforce_zero(uint32_t var) { row(var, 0, 0, 1); }
    uint32_t add(uint32_t a, uint32_t b) {
        const uint32_t c = push(hq_add(variables[a], variables[b]), mk_instr(W_ADD, a, b));
        row(a, b, c, 1);
        return c;
    }
    uint32_t mul(uint32_t a, uint32_t b) {
        const uint32_t c = push(hq_mul(variables[a], variables[b]), mk_instr(W_MUL, a, b));
        row(a, b, c, 0);
        return c;
    }
    uint32_t assemble_poseidon_gate(uint32_t a, uint32_t b) {
        const uint32_t c = push(hq_mul(variables[a], variables[b]), mk_instr(W_MUL, a, b));
        row(a, b, c, 0, c);
        return c;
    }
    // program_k: the constant the program uses where the reference's gate constant follows the witness (pm_select): the
    // gate row keeps the reference's constant for the template and is listed in witness_ops for every other proof
    uint32_t mul_constant(uint32_t a, uint32_t k, uint32_t program_k) {
        const uint32_t c = push(hq_scale(variables[a], k), mk_instr(W_MULC, a, 0, program_k));
        witness_ops.push_back(WitnessOp3{(uint32_t)rows.size(), a, program_k});
        row(a, 0, c, k);
        return c;
    }
    uint32_t mul_constant_plain(uint32_t a, uint32_t k) {
        const uint32_t c = push(hq_scale(variables[a], k), mk_instr(W_MULC, a, 0, k));
        row(a, 0, c, k);
        return c;
    }
    uint32_t mul_constant(uint32_t a, uint32_t k) { return mul_constant_plain(a, k); }
    Instr take_origin(Mode mode, const Q4& v) {
        if (mode == CONSTANT) return mk_instr(W_CONST, 0, 0, v[0], v[1], v[2], v[3]);
        if (!have_hint) throw std::logic_error("witness without provenance");
        have_hint = false;
        return hint;
    }
    uint32_t new_m31(uint32_t v, Mode mode) {
        const Q4 q{v, 0, 0, 0};
        const uint32_t c = push(q, take_origin(mode, q));
        if (mode == WITNESS) row(c, 0, c, 1, 0, 1);
        else row(1, 0, c, v);
        return c;
    }
    uint32_t new_qm31(const Q4& v, Mode mode) {
        const uint32_t c = push(v, take_origin(mode, v));
        if (mode == CONSTANT) {
            const uint32_t a0 = new_m31(v[0], CONSTANT), a1 = new_m31(v[1], CONSTANT), a2 = new_m31(v[2], CONSTANT), a3 = new_m31(v[3], CONSTANT);
            uint32_t t = mul(a1, 2);
            const uint32_t a = add(a0, t);
            t = mul(a3, 2);
            t = add(a2, t);
            const uint32_t b = mul(t, 3);
            row(a, b, c, 1);
        }
        return c;
    }
    void set_hint(const Instr& in) { hint = in; have_hint = true; }
};

// M31Var / CM31Var / QM31Var: a value and the index of its variable; kind = 1, 2 or 4 coordinates
struct Var {
    ConstraintSystem* cs;
    Q4 value;
    uint32_t variable;
    int kind;
};
inline Var mk(ConstraintSystem* cs, const Q4& v, uint32_t variable, int kind) { return Var{cs, v, variable, kind}; }

inline Var add(const Var& a0, const Var& b0) {
    const Var *a = &a0, *b = &b0;
    const int kind = a->kind > b->kind ? a->kind : b->kind;
    if (a->kind < b->kind) std::swap(a, b);  // `&M31Var + &QM31Var` and its likes are written `rhs + self`: the wires swap
    return mk(a->cs, hq_add(a->value, b->value), a->cs->add(a->variable, b->variable), kind);
}
inline Var neg(const Var& a) { return mk(a.cs, hq_neg(a.value), a.cs->mul_constant(a.variable, MP - 1), a.kind); }
inline Var sub(const Var& a, const Var& b) { return add(a, neg(b)); }  // self + &(-rhs)
inline Var mul(const Var& a0, const Var& b0) {
    const Var *a = &a0, *b = &b0;
    const int kind = a->kind > b->kind ? a->kind : b->kind;
    if (a->kind < b->kind) std::swap(a, b);
    return mk(a->cs, hq_mul(a->value, b->value), a->cs->mul(a->variable, b->variable), kind);
}
inline Var mul_constant(const Var& a, uint32_t k) { return mk(a.cs, hq_scale(a.value, k % MP), a.cs->mul_constant(a.variable, k % MP), a.kind); }
inline void equalverify(const Var& a, const Var& b) {
    if (a.value != b.value) throw std::runtime_error("equalverify: the template proof does not satisfy the circuit");
    a.cs->insert_gate(a.variable, 0, b.variable, 1);
}

inline Var m31_zero(ConstraintSystem* cs) { return mk(cs, {0, 0, 0, 0}, 0, 1); }
inline Var m31_one(ConstraintSystem* cs) { return mk(cs, {1, 0, 0, 0}, 1, 1); }
inline Var m31_constant(ConstraintSystem* cs, uint32_t v) {
    v %= MP;
    if (v == 0) return m31_zero(cs);
    if (v == 1) return m31_one(cs);
    const std::string key = "m31 " + std::to_string(v);
    auto it = cs->cache.find(key);
    if (it == cs->cache.end()) it = cs->cache.emplace(key, cs->new_m31(v, CONSTANT)).first;
    return mk(cs, {v, 0, 0, 0}, it->second, 1);
}
inline Var m31_witness(ConstraintSystem* cs, uint32_t v, const Instr& hint) {
    cs->set_hint(hint);
    return mk(cs, {v % MP, 0, 0, 0}, cs->new_m31(v % MP, WITNESS), 1);
}
inline Var m31_inv(const Var& a) {
    Var res = m31_witness(a.cs, h_inv(a.value[0]), mk_instr(W_INV, a.variable));
    a.cs->insert_gate(a.variable, res.variable, 1, 0);
    return res;
}
inline Var cm31_from_m31(const Var& real, const Var& imag) {
    ConstraintSystem* cs = real.cs;
    const uint32_t t = cs->mul(imag.variable, 2);
    return mk(cs, {real.value[0], imag.value[0], 0, 0}, cs->add(real.variable, t), 2);
}
inline Var cm31_inv(const Var& a) {  // no gate ties it to `a` (cm31.rs:240-246)
    const C2 v = hc_inv({a.value[0], a.value[1]});
    Var real = m31_witness(a.cs, v.re, mk_instr(W_CINV, a.variable, 0, 0));
    Var imag = m31_witness(a.cs, v.im, mk_instr(W_CINV, a.variable, 0, 1));
    return cm31_from_m31(real, imag);
}
inline Var shift_by_i(const Var& a) {
    const int kind = a.kind > 2 ? a.kind : 2;
    return mk(a.cs, hq_mul(a.value, {0, 1, 0, 0}), a.cs->mul(a.variable, 2), kind);
}
inline Var shift_by_j(const Var& a) { return mk(a.cs, hq_mul(a.value, {0, 0, 1, 0}), a.cs->mul(a.variable, 3), 4); }
inline Var shift_by_ij(const Var& a) { return shift_by_j(shift_by_i(a)); }

inline Var qm31_zero(ConstraintSystem* cs) { return mk(cs, {0, 0, 0, 0}, 0, 4); }
inline Var qm31_one(ConstraintSystem* cs) { return mk(cs, {1, 0, 0, 0}, 1, 4); }
inline Var qm31_witness(ConstraintSystem* cs, const Q4& v, const Instr& hint) {
    cs->set_hint(hint);
    return mk(cs, v, cs->new_qm31(v, WITNESS), 4);
}
inline Var qm31_constant(ConstraintSystem* cs, const Q4& v) {
    const Q4 fixed[4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    for (uint32_t k = 0; k < 4; k++)
        if (v == fixed[k]) return mk(cs, v, k, 4);
    const std::string key = "qm31 " + std::to_string(v[0]) + "," + std::to_string(v[1]) + "," + std::to_string(v[2]) + "," + std::to_string(v[3]);
    auto it = cs->cache.find(key);
    if (it == cs->cache.end()) it = cs->cache.emplace(key, cs->new_qm31(v, CONSTANT)).first;
    return mk(cs, v, it->second, 4);
}
inline Var qm31_from_m31(const Var& a0, const Var& a1, const Var& a2, const Var& a3) {
    ConstraintSystem* cs = a0.cs;
    const uint32_t l = cs->add(a0.variable, cs->mul(a1.variable, 2));
    const uint32_t r = cs->mul(cs->add(a2.variable, cs->mul(a3.variable, 2)), 3);
    return mk(cs, {a0.value[0], a1.value[0], a2.value[0], a3.value[0]}, cs->add(l, r), 4);
}
inline Var as_qm31(const Var& a) { return mk(a.cs, a.value, a.variable, 4); }
inline Var as_cm31(const Var& a) { return mk(a.cs, {a.value[0], a.value[1], 0, 0}, a.variable, 2); }
inline std::array<Var, 4> decompose_m31(const Var& a) {
    ConstraintSystem* cs = a.cs;
    std::array<Var, 4> parts{m31_witness(cs, a.value[0], mk_instr(W_COORD, a.variable, 0, 0)),
                             m31_witness(cs, a.value[1], mk_instr(W_COORD, a.variable, 0, 1)),
                             m31_witness(cs, a.value[2], mk_instr(W_COORD, a.variable, 0, 2)),
                             m31_witness(cs, a.value[3], mk_instr(W_COORD, a.variable, 0, 3))};
    const uint32_t l = cs->add(parts[0].variable, cs->mul(parts[1].variable, 2));
    const uint32_t r = cs->mul(cs->add(parts[2].variable, cs->mul(parts[3].variable, 2)), 3);
    cs->insert_gate(l, r, a.variable, 1);
    return parts;
}
inline std::array<Var, 2> decompose_cm31(const Var& a) {
    const auto v = decompose_m31(a);
    Var a0 = add(shift_by_i(as_cm31(v[1])), v[0]);
    Var a1 = add(shift_by_i(as_cm31(v[3])), v[2]);
    return {a0, a1};
}
inline Var qm31_inv(const Var& a) {
    Var res = qm31_witness(a.cs, hq_inv(a.value), mk_instr(W_QINV, a.variable));
    a.cs->insert_gate(a.variable, res.variable, 1, 0);
    return res;
}
inline std::pair<Var, Var> swap(const Var& a, const Var& b, bool bit_value, uint32_t bit_variable) {
    ConstraintSystem* cs = a.cs;
    const Var d = sub(b, a);
    uint32_t left = cs->mul(d.variable, bit_variable);
    uint32_t right = cs->mul_constant(left, MP - 1);
    left = cs->add(a.variable, left);
    right = cs->add(b.variable, right);
    return {mk(cs, bit_value ? b.value : a.value, left, a.kind), mk(cs, bit_value ? a.value : b.value, right, b.kind)};
}

}  // namespace rsv::circuit
