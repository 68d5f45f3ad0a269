// field.hpp — M31 / CM31 / QM31 arithmetic for gfx950 device code.
//
// Values side of primitives/fields/src/{m31,cm31,qm31}.rs (the reference keeps
// stwo's M31/CM31/QM31 inside M31Var/CM31Var/QM31Var; only the `value` half is
// reproduced here).  All inputs and outputs are canonical words in [0, P).
//
// CDNA4 notes: a 31x31->62 bit product is one v_mad_u64_u32 (measured 4.5-5 cycles per wave-instruction per SIMD on
// gfx950: the same class as v_min / v_alignbit, NOT quarter rate — poseidon2.hpp has the table); the Mersenne fold is
// v_alignbit + v_and + v_add, and the final conditional subtract is the branch-free `min(s, s - P)` (v_subrev +
// v_min_u32).  No divides, no MFMA (31-bit modular integer work).  These are the CANONICAL forms (inputs and outputs in
// [0, P)); the permutation's hot paths carry weakly reduced values instead (poseidon2.hpp, poseidon2_row.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsv {

constexpr uint32_t P = 0x7fffffffu;

__device__ __forceinline__ uint32_t m_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    return min(s, s - P);
}
__device__ __forceinline__ uint32_t m_sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return min(d, d + P);
}
__device__ __forceinline__ uint32_t m_neg(uint32_t a) { return m_sub(0u, a); }
__device__ __forceinline__ uint32_t m_dbl(uint32_t a) { return m_add(a, a); }
__device__ __forceinline__ uint32_t m_mul(uint32_t a, uint32_t b) {
    uint64_t p = (uint64_t)a * b;
    uint32_t s = ((uint32_t)p & P) + (uint32_t)(p >> 31);
    return min(s, s - P);
}
__device__ __forceinline__ uint32_t m_sqr(uint32_t a) { return m_mul(a, a); }
// x * 2^k mod P for canonical x, 0 < k < 31: a 31-bit rotate.
__device__ __forceinline__ uint32_t m_shl(uint32_t a, uint32_t k) {
    return ((a << k) & P) | (a >> (31u - k));
}
// a^(P-2) by the addition chain 2^31-3 = 2^31 - 1 - 2: 30 squarings + 7 products.
__device__ inline uint32_t m_inv(uint32_t a) {
    // t_k = a^(2^k - 1)
    uint32_t t1 = a;
    uint32_t t2 = m_mul(m_sqr(t1), t1);                      // 2^2-1
    uint32_t t4 = t2;
    for (int i = 0; i < 2; i++) t4 = m_sqr(t4);
    t4 = m_mul(t4, t2);                                      // 2^4-1
    uint32_t t8 = t4;
    for (int i = 0; i < 4; i++) t8 = m_sqr(t8);
    t8 = m_mul(t8, t4);                                      // 2^8-1
    uint32_t t16 = t8;
    for (int i = 0; i < 8; i++) t16 = m_sqr(t16);
    t16 = m_mul(t16, t8);                                    // 2^16-1
    uint32_t t24 = t16;
    for (int i = 0; i < 8; i++) t24 = m_sqr(t24);
    t24 = m_mul(t24, t8);                                    // 2^24-1
    uint32_t t28 = t24;
    for (int i = 0; i < 4; i++) t28 = m_sqr(t28);
    t28 = m_mul(t28, t4);                                    // 2^28-1
    uint32_t t29 = m_mul(m_sqr(t28), t1);                    // 2^29-1
    // a^(2^31-3) = (a^(2^29-1))^4 * a
    return m_mul(m_sqr(m_sqr(t29)), t1);
}

struct CM31 {
    uint32_t a, b;  // a + b*i
};
struct QM31 {
    CM31 a, b;  // a + b*u, u^2 = 2 + i
};

__device__ __forceinline__ CM31 c_mk(uint32_t a, uint32_t b) { return CM31{a, b}; }
__device__ __forceinline__ CM31 c_add(CM31 x, CM31 y) { return {m_add(x.a, y.a), m_add(x.b, y.b)}; }
__device__ __forceinline__ CM31 c_sub(CM31 x, CM31 y) { return {m_sub(x.a, y.a), m_sub(x.b, y.b)}; }
__device__ __forceinline__ CM31 c_neg(CM31 x) { return {m_neg(x.a), m_neg(x.b)}; }
__device__ __forceinline__ CM31 c_mul(CM31 x, CM31 y) {
    return {m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a))};
}
__device__ __forceinline__ CM31 c_mul_m(CM31 x, uint32_t k) { return {m_mul(x.a, k), m_mul(x.b, k)}; }
// x * (2 + i)
__device__ __forceinline__ CM31 c_mul_r(CM31 x) {
    return {m_sub(m_dbl(x.a), x.b), m_add(m_dbl(x.b), x.a)};
}
__device__ inline CM31 c_inv(CM31 x) {
    uint32_t n = m_inv(m_add(m_sqr(x.a), m_sqr(x.b)));
    return {m_mul(x.a, n), m_mul(m_neg(x.b), n)};
}

__device__ __forceinline__ QM31 q_mk(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
    return QM31{{a0, a1}, {b0, b1}};
}
__device__ __forceinline__ QM31 q_zero() { return q_mk(0, 0, 0, 0); }
__device__ __forceinline__ QM31 q_one() { return q_mk(1, 0, 0, 0); }
__device__ __forceinline__ QM31 q_from_m(uint32_t x) { return q_mk(x, 0, 0, 0); }
__device__ __forceinline__ QM31 q_add(QM31 x, QM31 y) { return {c_add(x.a, y.a), c_add(x.b, y.b)}; }
__device__ __forceinline__ QM31 q_sub(QM31 x, QM31 y) { return {c_sub(x.a, y.a), c_sub(x.b, y.b)}; }
__device__ __forceinline__ QM31 q_neg(QM31 x) { return {c_neg(x.a), c_neg(x.b)}; }
__device__ __forceinline__ QM31 q_dbl(QM31 x) { return q_add(x, x); }
__device__ __forceinline__ QM31 q_mul(QM31 x, QM31 y) {
    CM31 ac = c_mul(x.a, y.a), bd = c_mul(x.b, y.b);
    return {c_add(ac, c_mul_r(bd)), c_add(c_mul(x.a, y.b), c_mul(x.b, y.a))};
}
__device__ __forceinline__ QM31 q_mul_m(QM31 x, uint32_t k) { return {c_mul_m(x.a, k), c_mul_m(x.b, k)}; }
__device__ __forceinline__ QM31 q_mul_c(QM31 x, CM31 k) { return {c_mul(x.a, k), c_mul(x.b, k)}; }
__device__ inline QM31 q_inv(QM31 x) {
    CM31 den = c_inv(c_sub(c_mul(x.a, x.a), c_mul_r(c_mul(x.b, x.b))));
    return {c_mul(x.a, den), c_mul(c_neg(x.b), den)};
}
__device__ __forceinline__ bool q_eq(QM31 x, QM31 y) {
    return ((x.a.a ^ y.a.a) | (x.a.b ^ y.a.b) | (x.b.a ^ y.b.a) | (x.b.b ^ y.b.b)) == 0;
}
// shift_by_i / shift_by_j / shift_by_ij (primitives/fields/src/qm31.rs:402-418,466-468):
// multiplication by the basis elements is a signed permutation of the 4 words.
__device__ __forceinline__ QM31 q_mul_i(QM31 x) { return q_mk(m_neg(x.a.b), x.a.a, m_neg(x.b.b), x.b.a); }
// (a + b u) * u = b*(2+i) + a u
__device__ __forceinline__ QM31 q_mul_u(QM31 x) { return {c_mul_r(x.b), x.a}; }
// combine_ef (components/recursive/composition/src/data_structures.rs:142-145)
__device__ __forceinline__ QM31 q_combine_ef(QM31 v0, QM31 v1, QM31 v2, QM31 v3) {
    return q_add(q_add(v0, q_mul_i(v1)), q_add(q_mul_u(v2), q_mul_u(q_mul_i(v3))));
}

}  // namespace rsv
