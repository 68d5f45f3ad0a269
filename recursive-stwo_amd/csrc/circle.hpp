// circle.hpp — circle-group points over M31 needed by the verifier
// (values side of primitives/circle/src/lib.rs and of the stwo CanonicCoset /
// Coset::half_odds conventions listed in SURVEY App. B.2).
#pragma once
#include "field.hpp"

namespace rsv {

struct CPoint {
    uint32_t x, y;
};

namespace detail {
constexpr uint32_t cm_mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % 0x7fffffffu); }
constexpr uint32_t cm_add(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % 0x7fffffffu); }
constexpr uint32_t cm_sub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + 0x7fffffffu - b) % 0x7fffffffu); }
struct GenTable {
    uint32_t x[31], y[31];
};
// GEN * 2^i for the generator (2, 1268011823) of the order-2^31 circle group.
constexpr GenTable make_gen_table() {
    GenTable t{};
    uint32_t x = 2, y = 1268011823u;
    for (int i = 0; i < 31; i++) {
        t.x[i] = x;
        t.y[i] = y;
        uint32_t nx = cm_sub(cm_mul(x, x), cm_mul(y, y));
        uint32_t ny = cm_add(cm_mul(x, y), cm_mul(x, y));
        x = nx;
        y = ny;
    }
    return t;
}
}  // namespace detail

__constant__ detail::GenTable GEN_POW = detail::make_gen_table();

__device__ __forceinline__ CPoint cp_add(CPoint p, CPoint q) {
    return {m_sub(m_mul(p.x, q.x), m_mul(p.y, q.y)), m_add(m_mul(p.x, q.y), m_mul(p.y, q.x))};
}
// k * GEN, k taken mod 2^31
__device__ inline CPoint cp_gen_mul(uint32_t k) {
    CPoint acc = {1u, 0u};
#pragma unroll 1
    for (int i = 0; i < 31; i++) {
        CPoint g = {GEN_POW.x[i], GEN_POW.y[i]};
        CPoint s = cp_add(acc, g);
        bool bit = (k >> i) & 1u;
        acc.x = bit ? s.x : acc.x;
        acc.y = bit ? s.y : acc.y;
    }
    return acc;
}
// Coset::half_odds(n).at(i) = g_{n+2} + i * g_n with g_n = GEN * 2^(31-n)
__device__ inline CPoint half_odds_at(uint32_t n, uint32_t i) {
    uint32_t k = ((1u << (29u - n)) + (i << (31u - n))) & P;
    return cp_gen_mul(k);
}
__device__ __forceinline__ uint32_t bit_reverse(uint32_t v, uint32_t bits) { return __brev(v) >> (32u - bits); }
// CanonicCoset(log).circle_domain().at(bit_reverse(q, log)): the point carried by
// PointCarryingQueryVar (primitives/query/src/lib.rs:57-143).
__device__ inline CPoint domain_point(uint32_t log_size, uint32_t q) {
    uint32_t i = bit_reverse(q, log_size), half = 1u << (log_size - 1u);
    CPoint p = half_odds_at(log_size - 1u, i & (half - 1u));
    if (i & half) p.y = m_neg(p.y);
    return p;
}

}  // namespace rsv
