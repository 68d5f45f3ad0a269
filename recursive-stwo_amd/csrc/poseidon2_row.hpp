// poseidon2_row.hpp — Poseidon2-M31 with ONE STATE PER 16-LANE DPP ROW (4 permutations per wave64).
//
// This is the latency form of the permutation: lane i of a row holds state word i, the S-boxes of a full
// round run in all 16 lanes at once, the external matrix is three quad rotations (row k of
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] read from x_k onwards is (5,7,1,3) for even k and (6,1,1,4)
// for odd k) plus row rotations by 4/8/12 (sum over the quads), and the partial-round sum is a 4-step
// rotate-and-add all-reduce.  One permutation is ~1.1 k
// wave-instructions of latency instead of ~4.6 k, but a wave only carries 4 of them, so it costs ~4x more
// issue slots per permutation than the lane-per-state form (poseidon2.hpp).  It is used where the work is a
// strictly sequential chain and there are too few proofs to fill the machine with independent lanes: the
// Fiat-Shamir transcript of small batches (k_transcript_row).
//
// Values side of primitives/poseidon31/src/implementation.rs:7-149; all words canonical.
#pragma once
#include "poseidon2.hpp"

namespace rsv {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
// quad_perm:[1,2,3,0] / [2,3,0,1] / [3,0,1,2]; row_ror:n = 0x120 + n
constexpr int DPP_QROT1 = 0x39, DPP_QROT2 = 0x4E, DPP_QROT3 = 0x93;
constexpr int DPP_ROR1 = 0x121, DPP_ROR2 = 0x122, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128, DPP_ROR12 = 0x12C;

// external matrix circ(2*M4, M4, M4, M4) on a row; odd = lane & 1
__device__ __forceinline__ uint32_t mds_row(uint32_t x, bool odd) {
    const uint32_t ca = odd ? 6u : 5u, cb = odd ? 1u : 7u, cd = odd ? 4u : 3u;
    uint32_t b = dpp<DPP_QROT1>(x), c = dpp<DPP_QROT2>(x), d = dpp<DPP_QROT3>(x);
    uint64_t acc = (uint64_t)x * ca + c;                // < 2^34
    acc = (uint64_t)b * cb + acc;
    acc = (uint64_t)d * cd + acc;                       // < 16 * 2^31
    uint32_t t = ((uint32_t)acc & P) + (uint32_t)(acc >> 31);  // <= P + 16
    uint32_t y = min(t, t - P);
    uint32_t col = m_add(m_add(y, dpp<DPP_ROR4>(y)), m_add(dpp<DPP_ROR8>(y), dpp<DPP_ROR12>(y)));
    return m_add(y, col);
}
__device__ __forceinline__ uint32_t sum_row(uint32_t x) {
    x = m_add(x, dpp<DPP_ROR1>(x));
    x = m_add(x, dpp<DPP_ROR2>(x));
    x = m_add(x, dpp<DPP_ROR4>(x));
    return m_add(x, dpp<DPP_ROR8>(x));
}

// (RowRC / load_row_rc: the lane's round constants, fetched once per kernel — poseidon2.hpp)

// row_newbcast:0 — lane 0 of every row, in all 16 lanes of that row
constexpr int DPP_BCAST0 = 0x150;

// ===========================================================================================================
// The permutation with FEWER INSTRUCTIONS (round 4).  A small batch's transcript is 233 permutations one after the
// other on one wave per SIMD.  A wave issues in order, about one instruction per 4.3 cycles whatever the dependencies
// (measured: rounds 1-3's form, 1 563 issue slots per permutation, took 2.8 us), so what such a chain costs is its
// instruction COUNT.  Rounds 1-3 kept every intermediate canonical (m_add / m_mul: 3 and 6 instructions each).  Here values
// are only congruent mod P and held to a proven range, as in the lane form (poseidon2.hpp);
// tests/test_perm_arithmetic.py::test_row_form_weak_range_model is this code statement by statement in
// Python integers with every width and range asserted:
//   mulw      any u32 x any u32 -> <= 2 P + 3 in 6 instructions: the 64-bit product as three 31-bit limbs added up
//             (2^31 = 1 mod P); nothing is canonicalised between the three products of x^5.
//   mds_row_w the M4 products of a quad accumulate in 64 bits (< 2^36); the sum over the four quads is taken ON THE
//             64-BIT accumulators with v_add_co_u32_dpp / v_addc_co_u32_dpp (the rotation rides on the add: no moves),
//             then ONE Mersenne fold that also adds the next layer's round constant (v_add3).
//   partial   every lane computes the S-box of word 0.  The sum of words 1..15 is a 64-bit rotate-and-add all-reduce
//             (exact: 15 terms < 2^32), and since every lane then holds the sum AND the S-box output, every lane also
//             computes word 0's NEXT S-box input (4 s + rest + rc) itself: no broadcast, and lane 0's own register is
//             only written behind the last round.  A lane's own word is d_i x_i + rest + s < 2^50 -> fold <= P + 2^19.
// Every round constant is < P - 2^19 (static_assert in poseidon2.hpp), so x + rc never wraps.
// DPP hazard: a VGPR written by a VALU instruction may be read through DPP two wait states later at the earliest; the
// asm blocks below carry their own s_nop (the assembler adds none), the compiler handles its own DPP moves.

// launder: the value is what it is, but the compiler forgets what it knew about its range.  (Knowing that a 32-bit sum
// cannot wrap, LLVM widens it to 64 bits to absorb the zero-extension of the next product and multiplies 64 x 32.)
__device__ __forceinline__ uint32_t opaque_v(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
// any u32 x any u32 -> congruent, <= 2 P + 3.  t = A 2^62 + B 2^31 + C = A + B + C (mod P)
__device__ __forceinline__ uint32_t mulw(uint32_t a, uint32_t b) {
    const uint64_t t = (uint64_t)a * b;
    const uint32_t hi = (uint32_t)(t >> 32), lo = (uint32_t)t;
    const uint32_t B = __builtin_amdgcn_alignbit(hi, lo, 31) & P;
    return opaque_v((lo & P) + B + (hi >> 30));
}
__device__ __forceinline__ uint32_t sbox_w(uint32_t u) {
    const uint32_t x2 = mulw(u, u);
    return mulw(mulw(x2, x2), u);
}
// v < 2^50 -> (v & P) + (v >> 31) + add: congruent to v + add, fits 32 bits for the ranges stated at the call sites
__device__ __forceinline__ uint32_t foldw(uint64_t v, uint32_t add = 0) {
    return opaque_v(((uint32_t)v & P) + (uint32_t)(v >> 31) + add);
}
// bound_ctrl set and every lane a valid source: the old value of the destination is never used (no zeroing move)
template <int CTRL>
__device__ __forceinline__ uint32_t dppb(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
#define RSV_DPP_ADD64(ROR) \
    "v_add_co_u32_dpp %0, vcc, %2, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t" \
    "v_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
// w += ror4(a) + ror8(a) + ror12(a), all 64-bit: the sum over the four quads of a row
__device__ __forceinline__ uint64_t add_quads64(uint64_t w, uint64_t a) {
    uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
    const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
    asm("s_nop 1\n\t" RSV_DPP_ADD64(4) RSV_DPP_ADD64(8) RSV_DPP_ADD64(12)
        : "+v"(wlo), "+v"(whi) : "v"(alo), "v"(ahi) : "vcc");
    return ((uint64_t)whi << 32) | wlo;
}
#define RSV_DPP_ACC64(ROR) \
    "v_add_co_u32_dpp %0, vcc, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t" \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
// sum over the 16 lanes of a row of 32-bit values, exact in 64 bits, in every lane
__device__ __forceinline__ uint64_t sum_row64(uint32_t x) {
    uint32_t lo = x, hi = 0;
    asm("s_nop 1\n\t" RSV_DPP_ACC64(1) "s_nop 0\n\t" RSV_DPP_ACC64(2) "s_nop 0\n\t" RSV_DPP_ACC64(4) "s_nop 0\n\t" RSV_DPP_ACC64(8)
        : "+v"(lo), "+v"(hi) : : "vcc");
    return ((uint64_t)hi << 32) | lo;
}
// external matrix on a row: x any u32 -> (circ(2 M4, M4, M4, M4) x)_i + add, weak (<= P + 2^8 + add)
__device__ __forceinline__ uint32_t mds_row_w(uint32_t x, bool odd, uint32_t add) {
    const uint32_t ca = odd ? 6u : 5u, cb = odd ? 1u : 7u, cd = odd ? 4u : 3u;
    const uint32_t b = dppb<DPP_QROT1>(x), c = dppb<DPP_QROT2>(x), d = dppb<DPP_QROT3>(x);
    uint64_t acc = (uint64_t)x * ca + c;
    acc = (uint64_t)b * cb + acc;
    acc = (uint64_t)d * cd + acc;                                             // < 2^36
    return foldw(add_quads64(acc << 1, acc), add);                            // 2 acc + the other three quads < 2^39
}

// One partial round.  x: the lane's word (any u32; lane 0's is not used); u: the S-box input of word 0 (x_0 + rc), the
// same in every lane.  Returns the lane's new word (lane 0: not meaningful); u becomes the next round's S-box input.
template <uint32_t NEXT_RC>
__device__ __forceinline__ uint32_t partial_round_w(uint32_t x, uint32_t i, uint32_t diag, uint32_t& u) {
    const uint32_t xm = i == 0 ? 0u : x;
    const uint64_t rest = sum_row64(xm);                                      // words 1..15, exact, < 2^36
    const uint64_t pre = (uint64_t)xm * diag + rest;                          // lane i >= 1: d_i x_i + rest (< 2^49)
    const uint32_t s = sbox_w(u);                                             // S-box of word 0, in every lane
    u = foldw((uint64_t)s * 4u + rest, NEXT_RC);                              // word 0's new value 3 s + (s + rest), + next rc
    return foldw(pre + s);                                                    // d_i x_i + (s + rest) <= P + 2^19
}

// x: state word (lane & 15) of this row's state (any u32: a non-canonical proof word is permuted as the field element
// it is congruent to).  i = lane & 15.  Returns the canonical output word.
__device__ __forceinline__ uint32_t poseidon2_row(uint32_t x, uint32_t i, const RowRC& k) {
    const uint32_t diag = opaque_v(1u << (i + 1));  // (lane 0's is not used: its xm is 0; opaque: a multiplier, not a 64-bit variable shift)
    const bool odd = i & 1u;
    x = mds_row_w(x, odd, k.f[0]);
#pragma unroll
    for (int r = 0; r < 3; r++) x = mds_row_w(sbox_w(x), odd, k.f[r + 1]);
    x = mds_row_w(sbox_w(x), odd, 0u);
    uint32_t u = dppb<DPP_BCAST0>(x) + RC_PARTIAL_K[0];
    x = partial_round_w<RC_PARTIAL_K[1]>(x, i, diag, u);   x = partial_round_w<RC_PARTIAL_K[2]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[3]>(x, i, diag, u);   x = partial_round_w<RC_PARTIAL_K[4]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[5]>(x, i, diag, u);   x = partial_round_w<RC_PARTIAL_K[6]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[7]>(x, i, diag, u);   x = partial_round_w<RC_PARTIAL_K[8]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[9]>(x, i, diag, u);   x = partial_round_w<RC_PARTIAL_K[10]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[11]>(x, i, diag, u);  x = partial_round_w<RC_PARTIAL_K[12]>(x, i, diag, u);
    x = partial_round_w<RC_PARTIAL_K[13]>(x, i, diag, u);  x = partial_round_w<0u>(x, i, diag, u);
    x = (i == 0 ? u : x) + k.f[4];                                            // word 0 is u (no rc behind the last round); <= P + 2^19 + rc < 2^32
#pragma unroll
    for (int r = 4; r < 7; r++) x = mds_row_w(sbox_w(x), odd, k.f[r + 1]);
    x = mds_row_w(sbox_w(x), odd, 0u);                                        // <= P + 2^8
    return min(x, x - P);
}

// row_share:N — lane N of every row, in all 16 lanes of that row
template <int N>
__device__ __forceinline__ uint32_t row_share(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, DPP_BCAST0 + N, 0xF, 0xF, true);
}

// lanes whose bit in `lanes` is set take b, the others a
__device__ __forceinline__ uint32_t lane_select(uint32_t a, uint32_t b, uint64_t lanes) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(lanes));
    return r;
}

// the row's permutation of a state that every lane of the row holds: lane i takes word i and returns output word i
__device__ __forceinline__ uint32_t poseidon2_row_of_state(const State16& st) {
    const uint32_t i = threadIdx.x & 15u;  // (workgroups are multiples of 64 threads: rows are aligned)
    RowRC k;                               // issued first: the first constant is needed ~45 instructions on
#pragma unroll
    for (int r = 0; r < 8; r++) k.f[r] = s_row_rc[r][i];
    // (asm: written as ?: the compiler sees sixteen values picked by an index, stores the state to scratch memory and
    // loads one word back — a memory round trip per permutation and scratch in every kernel that calls this)
    uint32_t a[8], b[4], c[2];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = lane_select(st.s[2 * j], st.s[2 * j + 1], 0xAAAAAAAAAAAAAAAAull);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = lane_select(a[2 * j], a[2 * j + 1], 0xCCCCCCCCCCCCCCCCull);
#pragma unroll
    for (int j = 0; j < 2; j++) c[j] = lane_select(b[2 * j], b[2 * j + 1], 0xF0F0F0F0F0F0F0F0ull);
    return poseidon2_row(lane_select(c[0], c[1], 0xFF00FF00FF00FF00ull), i, k);
}

// poseidon2_half on a "virtual lane" (poseidon2.hpp): every lane of the row passes the same state and receives the same
// half of the output.  Lane i takes word i of the state (four levels of selects on the bits of i), the row permutes, and
// the eight output words are read back from the lanes that hold them (row_share).  The callers' control flow is uniform
// over a row (all 16 lanes carry the same indices), so the row's lanes are active together wherever this is called.
template <bool HI>
__device__ __noinline__ Hash8 poseidon2_row_half(State16 st) {
#ifdef RSV_COUNT_PERMS
    {
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) {
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u)], (unsigned long long)__builtin_popcountll(m) / 16ull);
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u) + 1], 1ull);
        }
    }
#endif
    const uint32_t y = poseidon2_row_of_state(st);
    Hash8 h;
    h.w[0] = row_share<(HI ? 8 : 0) + 0>(y); h.w[1] = row_share<(HI ? 8 : 0) + 1>(y);
    h.w[2] = row_share<(HI ? 8 : 0) + 2>(y); h.w[3] = row_share<(HI ? 8 : 0) + 3>(y);
    h.w[4] = row_share<(HI ? 8 : 0) + 4>(y); h.w[5] = row_share<(HI ? 8 : 0) + 5>(y);
    h.w[6] = row_share<(HI ? 8 : 0) + 6>(y); h.w[7] = row_share<(HI ? 8 : 0) + 7>(y);
    return h;
}

// the whole output state in every lane of the row (the PoseidonFlow kernels' records hold all sixteen words)
__device__ __noinline__ State16 poseidon2_row_state(State16 st) {
#ifdef RSV_COUNT_PERMS
    {
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) {
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u)], (unsigned long long)__builtin_popcountll(m) / 16ull);
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u) + 1], 1ull);
        }
    }
#endif
    const uint32_t y = poseidon2_row_of_state(st);
    State16 o;
    o.s[0] = row_share<0>(y); o.s[1] = row_share<1>(y); o.s[2] = row_share<2>(y); o.s[3] = row_share<3>(y);
    o.s[4] = row_share<4>(y); o.s[5] = row_share<5>(y); o.s[6] = row_share<6>(y); o.s[7] = row_share<7>(y);
    o.s[8] = row_share<8>(y); o.s[9] = row_share<9>(y); o.s[10] = row_share<10>(y); o.s[11] = row_share<11>(y);
    o.s[12] = row_share<12>(y); o.s[13] = row_share<13>(y); o.s[14] = row_share<14>(y); o.s[15] = row_share<15>(y);
    return o;
}

}  // namespace rsv
