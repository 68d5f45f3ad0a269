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

// Round constants of one lane: RC_FULL[r][i] for the eight full rounds, fetched ONCE per kernel (a load inside the
// round loop is consumed two instructions later and costs the wave its whole latency, eight times per permutation —
// more than the arithmetic of the round).  The partial-round constants are literals.
struct RowRC {
    uint32_t f[8];
};
__device__ __forceinline__ RowRC load_row_rc(uint32_t i) {
    RowRC k;
#pragma unroll
    for (int r = 0; r < 8; r++) k.f[r] = RC_FULL[r][i];
    return k;
}

// row_newbcast:0 — lane 0 of every row, in all 16 lanes of that row
constexpr int DPP_BCAST0 = 0x150;

// One partial round.  Only word 0 goes through the S-box, and the sum of the other fifteen does not depend on it:
// every lane computes the S-box of the row's word 0 (broadcast), while the row all-reduce of words 1..15 and the
// diagonal products run in its shadow; what is left behind the S-box is a shift, a select and one modular add.
template <int R>
__device__ __forceinline__ uint32_t partial_round_row(uint32_t x, uint32_t i, uint32_t diag) {
    const uint32_t x0 = dpp<DPP_BCAST0>(x);
    const uint32_t rest = sum_row(i == 0 ? 0u : x);              // words 1..15
    const uint32_t q = m_add(m_mul(x, diag), rest);               // lanes 1..15: d_i * x_i + rest
    const uint32_t s = pow5_ref(m_add(x0, RC_PARTIAL_K[R]));      // S-box of word 0, the same in every lane
    // lane 0: 3 s + (s + rest) = 4 s + rest;  lane i: d_i x_i + (s + rest)
    return m_add(i == 0 ? rest : q, i == 0 ? m_shl(s, 2) : s);
}

// x: state word (lane & 15) of this row's state.  i = lane & 15.
__device__ __forceinline__ uint32_t poseidon2_row(uint32_t x, uint32_t i, const RowRC& k) {
    const uint32_t diag = i == 0 ? 3u : (1u << (i + 1));
    const bool odd = i & 1u;
    x = mds_row(x, odd);
#pragma unroll
    for (int r = 0; r < 4; r++) x = mds_row(pow5_ref(m_add(x, k.f[r])), odd);
    x = partial_round_row<0>(x, i, diag);   x = partial_round_row<1>(x, i, diag);
    x = partial_round_row<2>(x, i, diag);   x = partial_round_row<3>(x, i, diag);
    x = partial_round_row<4>(x, i, diag);   x = partial_round_row<5>(x, i, diag);
    x = partial_round_row<6>(x, i, diag);   x = partial_round_row<7>(x, i, diag);
    x = partial_round_row<8>(x, i, diag);   x = partial_round_row<9>(x, i, diag);
    x = partial_round_row<10>(x, i, diag);  x = partial_round_row<11>(x, i, diag);
    x = partial_round_row<12>(x, i, diag);  x = partial_round_row<13>(x, i, diag);
#pragma unroll
    for (int r = 4; r < 8; r++) x = mds_row(pow5_ref(m_add(x, k.f[r])), odd);
    return x;
}

}  // namespace rsv
