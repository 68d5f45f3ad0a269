// poseidon2.hpp — Poseidon2 over M31, width 16, x^5, 4 + 14 + 4 rounds.
//
// Values side of poseidon2_permute (primitives/poseidon31/src/implementation.rs:108-149)
// with the external matrix circ(2*M4, M4, M4, M4) (:7-58) and the internal
// matrix diag(3,4,8,...,65536) + J (:128-138).  The 142 round constants are the
// Poseidon2 parameters for p = 2^31-1, t = 16 published in
// primitives/poseidon31/src/parameters.rs:6-190.
//
// MI355X mapping: ONE LANE PER PERMUTATION.  The 16-word state lives in 16
// VGPRs of one lane, so a wave64 advances 64 independent permutations with no
// cross-lane traffic.  The fast path (poseidon2_inline, below) is straight-line
// code: round constants are 32-bit literals of a fused add + canonicalise, the
// linear layers accumulate unreduced in 64 bits, the partial-round diagonal is
// a multiply-accumulate by a power of two.  One out-of-line instance
// (poseidon2(), ~30 KB of code in 38 VGPRs) is shared by every call site.
#pragma once
#include "field.hpp"

namespace rsv {

struct State16 {
    uint32_t s[16];
};

#define RSV_RC_FULL_INIT { \
    {0x768bab52, 0x70e0ab7d, 0x3d266c8a, 0x6da42045, 0x600fef22, 0x41dace6b, 0x64f9bdd4, 0x5d42d4fe, \
     0x76b1516d, 0x6fc9a717, 0x70ac4fb6, 0x00194ef6, 0x22b644e2, 0x1f7916d5, 0x47581be2, 0x2710a123}, \
    {0x6284e867, 0x018d3afe, 0x5df99ef3, 0x4c1e467b, 0x566f6abc, 0x2994e427, 0x538a6d42, 0x5d7bf2cf, \
     0x7fda2dab, 0x0fd854c4, 0x46922fca, 0x3d7763a1, 0x19fd05ca, 0x0a4bbb43, 0x15075851, 0x3d903d76}, \
    {0x2d290ff7, 0x40809fa0, 0x59dac6ec, 0x127927a2, 0x6bbf0ea0, 0x0294140f, 0x24742976, 0x6e84c081, \
     0x22484f4a, 0x354cae59, 0x0453ffe1, 0x3f47a3cc, 0x0088204e, 0x6066e109, 0x3b7c4b80, 0x6b55665d}, \
    {0x3bc4b897, 0x735bf378, 0x508daf42, 0x1884fc2b, 0x7214f24c, 0x7498be0a, 0x1a60e640, 0x3303f928, \
     0x29b46376, 0x5c96bb68, 0x65d097a5, 0x1d358e9f, 0x4a9a9017, 0x4724cf76, 0x347af70f, 0x1e77e59a}, \
    {0x57090613, 0x1fa42108, 0x17bbef50, 0x1ff7e11c, 0x047b24ca, 0x4e140275, 0x4fa086f5, 0x079b309c, \
     0x1159bd47, 0x6d37e4e5, 0x075d8dce, 0x12121ca0, 0x7f6a7c40, 0x68e182ba, 0x5493201b, 0x0444a80e}, \
    {0x0064f4c6, 0x6467abe6, 0x66975762, 0x2af68f9b, 0x345b33be, 0x1b70d47f, 0x053db717, 0x381189cb, \
     0x43b915f8, 0x20df3694, 0x0f459d26, 0x77a0e97b, 0x2f73e739, 0x1876c2f9, 0x65a0e29a, 0x4cabefbe}, \
    {0x5abd1268, 0x4d34a760, 0x12771799, 0x69a0c9ac, 0x39091e55, 0x7f611cd0, 0x3af055da, 0x7ac0bbdf, \
     0x6e0f3a24, 0x41e3b6f7, 0x49b3756d, 0x568bc538, 0x20c079d8, 0x1701c72c, 0x7670dc6c, 0x5a439035}, \
    {0x7c93e00e, 0x561fbb4d, 0x1178907b, 0x02737406, 0x32fb24f1, 0x6323b60a, 0x6ab12418, 0x42c99cea, \
     0x155a0b97, 0x53d1c6aa, 0x2bd20347, 0x279b3d73, 0x4f5f3c70, 0x0245af6c, 0x238359d3, 0x49966a59}}
__constant__ __attribute__((aligned(64))) uint32_t RC_FULL[8][16] = RSV_RC_FULL_INIT;
constexpr uint32_t RC_FULL_K[8][16] = RSV_RC_FULL_INIT;  // the same values as compile-time constants (literal operands)

// 14 constants + 2 words of padding (block loads)
#define RSV_RC_PARTIAL_INIT { \
   0x7f7ec4bf, 0x0421926f, 0x5198e669, 0x34db3148, 0x4368bafd, \
                                        0x66685c7f, 0x78d3249a, 0x60187881, 0x76dad67a, 0x0690b437, \
                                        0x1ea95311, 0x40e5369a, 0x38f103fc, 0x1d226a21, 0, 0}
__constant__ __attribute__((aligned(64))) uint32_t RC_PARTIAL[16] = RSV_RC_PARTIAL_INIT;
constexpr uint32_t RC_PARTIAL_K[16] = RSV_RC_PARTIAL_INIT;

// ---- straightforward canonical implementation (readable restatement; baseline of tools/perm_lab.hip)
__device__ __forceinline__ uint32_t pow5_ref(uint32_t x) {
    uint32_t x2 = m_sqr(x);
    return m_mul(m_sqr(x2), x);
}

// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] (Poseidon2 paper, 5.1) in 8 additions, 2 doublings, 2 x4.
__device__ __forceinline__ void mds4_ref(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    uint32_t t0 = m_add(x0, x1), t1 = m_add(x2, x3);
    uint32_t t2 = m_add(m_dbl(x1), t1), t3 = m_add(m_dbl(x3), t0);
    uint32_t t4 = m_add(m_shl(t1, 2), t3), t5 = m_add(m_shl(t0, 2), t2);
    x0 = m_add(t3, t5);
    x1 = t5;
    x2 = m_add(t2, t4);
    x3 = t4;
}

__device__ __forceinline__ void mds16_ref(uint32_t* s) {
#pragma unroll
    for (int g = 0; g < 4; g++) mds4_ref(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t sum = m_add(m_add(s[j], s[j + 4]), m_add(s[j + 8], s[j + 12]));
#pragma unroll
        for (int g = 0; g < 4; g++) s[4 * g + j] = m_add(s[4 * g + j], sum);
    }
}

__device__ __forceinline__ void poseidon2_ref_inline(uint32_t* s) {
    mds16_ref(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = pow5_ref(m_add(s[i], RC_FULL[r][i]));
        mds16_ref(s);
    }
#pragma unroll 1
    for (int r = 0; r < 14; r++) {
        s[0] = pow5_ref(m_add(s[0], RC_PARTIAL[r]));
        // sum of all 16 words as a balanced tree
        uint32_t a0 = m_add(s[0], s[1]), a1 = m_add(s[2], s[3]), a2 = m_add(s[4], s[5]), a3 = m_add(s[6], s[7]);
        uint32_t a4 = m_add(s[8], s[9]), a5 = m_add(s[10], s[11]), a6 = m_add(s[12], s[13]),
                 a7 = m_add(s[14], s[15]);
        uint32_t sum = m_add(m_add(m_add(a0, a1), m_add(a2, a3)), m_add(m_add(a4, a5), m_add(a6, a7)));
        s[0] = m_add(sum, m_add(m_dbl(s[0]), s[0]));  // diag 3
#pragma unroll
        for (int i = 1; i < 16; i++) s[i] = m_add(sum, m_shl(s[i], i + 1));  // diag 2^(i+1)
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = pow5_ref(m_add(s[i], RC_FULL[r][i]));
        mds16_ref(s);
    }
}


// ===========================================================================================
// Fast path.  Instruction costs measured on MI355X (tools/valu_lab.hip, 4 waves/SIMD):
//   ~2.5 cycles : v_add_u32 v_sub_u32 v_and_b32 v_or_b32 v_lshrrev_b32 v_mov_b32 (also with a literal)
//   ~4.3-4.6    : v_min_u32 v_lshlrev_b32 v_alignbit_b32 v_and_or_b32 v_mul_lo/hi_u32 v_lshl_add_u64
//                 v_mad_u64_u32 (4.55; 5.05 with a live 64-bit addend)   -- the 32x32->64 multiply is
//                 NOT quarter rate on gfx950, so the cost of a modular multiply is its reduction.
// Consequences used below:
//   * linear layers accumulate UNREDUCED in 64 bits (one v_lshl_add_u64 / v_mad_u64_u32 per term,
//     shifts by 1..4 and small multipliers are free), and are folded once per round;
//   * accumulators hold 2*v, so the Mersenne fold (v >> 31) + (v & P) is hi32 + (lo32 >> 1): two
//     fast-class instructions instead of v_and + v_alignbit + v_add;
//   * a product x*y is formed as (2x)*y for the same reason;
//   * values between steps are only "weakly" reduced; the ranges are tracked in the comments:
//       C  = [0, P]           (canonical, or P itself which is congruent to 0)
//       L2 = [0, 2P]          (one conditional subtract away from C)
//     Bit-exactness with the canonical reference (poseidon2_ref_inline) is tested on the GPU.
// ===========================================================================================
// Issue pacing.  Measured on MI355X (whole pipeline, 65 536 proofs): when a wave presents an instruction that depends
// on its own previous VALU result, the SIMD stalls on it instead of issuing another wave's instruction.  One extra
// wait state behind every v_mad_u64_u32 (the compiler adds one of its own behind an asm statement whose result is
// read next) and two behind the v_min that ends a reduction take the wave out of arbitration for those cycles:
//     no pacing 37.75 ms | mad 36.40 | mad + canon 35.80 | also behind fold2 / the doublings: 35.75-35.85 (plateau)
//     two wait states behind the multiplies: 36.17 (worse) | s_nop 3: 40.2
// The wait states cost nothing at >= 4 waves per SIMD (other waves fill them); at one wave per SIMD (the lane-form
// transcript of batches > 24 576) they lengthen the chain by ~25 % — that kernel runs underneath k_row_hash.
// PACE: the wait states above (one behind every v_mad_u64_u32, one behind the v_min that ends a reduction).  They pay where
// several waves share a SIMD and cost ~25 % where a wave is (nearly) alone on it, so the verify kernels pick the instance by
// the size of the launch (poseidon2_half below): paced for launches that fill the machine, unpaced for a small batch's trees
// and for the one-wave-per-SIMD lane-form transcript.
template <bool PACE>
struct PermT {
    static __device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b) {
        uint64_t d;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    template <int SH>
    static __device__ __forceinline__ uint64_t shl_add64(uint64_t a, uint64_t b) {  // (a << SH) + b, SH in 1..4
        uint64_t d;
        asm("v_lshl_add_u64 %0, %1, %3, %2" : "=v"(d) : "v"(a), "v"(b), "n"(SH));
        return d;
    }
    // 32x32 -> 64 products and multiply-accumulates (v_mad_u64_u32).  The small constant multipliers are
    // passed as OPAQUE wave-uniform values (see opaque()): with a visible constant hipcc strength-reduces
    // a*2+c into slow-class shifts plus zero-extension moves instead of one v_mad_u64_u32.
    static __device__ __forceinline__ uint32_t opaque(uint32_t k) {
        uint32_t r;
        asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "n"(k));
        return r;
    }
    // The asm form (instead of `(uint64_t)a * b + c`) also keeps hipcc from re-associating
    // x0*k + x1*k into (x0 + x1)*k, which costs a 64-bit add, a 64x32 multiply and zero-extension moves.
    static __device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) {
        uint64_t d, carry;
        if constexpr (PACE) asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\ts_nop 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
        else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
        return d;
    }
    static __device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b_uniform, int) {  // b in an SGPR
        uint64_t d, carry;
        if constexpr (PACE) asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\ts_nop 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_uniform));
        else asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_uniform));
        return d;
    }
    static __device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b_uniform, uint64_t c) {  // a * b + c, b in an SGPR
        uint64_t d, carry;
        if constexpr (PACE) asm("v_mad_u64_u32 %0, %1, %2, %3, %4\n\ts_nop 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_uniform), "v"(c));
        else asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_uniform), "v"(c));
        return d;
    }
    static __device__ __forceinline__ uint32_t dbl32(uint32_t x) {  // x + x as a fast-class add (not a shift)
        uint32_t d;
        asm("v_add_u32 %0, %1, %1" : "=v"(d) : "v"(x));
        return d;
    }
    // V = 2v with v < 2^62  ->  (v >> 31) + (v & P)
    static __device__ __forceinline__ uint32_t fold2(uint64_t V) {
        uint32_t r = (uint32_t)(V >> 32) + ((uint32_t)V >> 1);
        return r;
    }
    // t in [0, 2P] -> C
    static __device__ __forceinline__ uint32_t canon(uint32_t t) {
        uint32_t r = min(t, t - P);
        if constexpr (PACE) asm volatile("s_nop 0" : "+v"(r));
        return r;
    }

    // x in C  ->  x^5 in L2
    static __device__ __forceinline__ uint32_t pow5(uint32_t x) {
        const uint32_t xx = dbl32(x);                         // 2x <= 2P, used by the first and the last product
        uint32_t c2 = canon(fold2(mul64(xx, x)));             // 2x^2 < 2^63; fold <= 2P-1
        uint32_t c4 = canon(fold2(mul64(dbl32(c2), c2)));
        return fold2(mul64(xx, c4));                          // 2x*c4: hi <= P, lo>>1 <= P
    }

    // Y = 2*M4*(x0..x3) for 32-bit inputs (any u32), exact in 64 bits.
    static __device__ __forceinline__ void mds4_2x(uint32_t k2, uint32_t k4, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3,
                                            uint64_t& y0, uint64_t& y1, uint64_t& y2, uint64_t& y3) {
        uint64_t T0 = mad64(x0, k2, mul64(x1, k2, 0));           // 2(x0 + x1)
        uint64_t T1 = mad64(x2, k2, mul64(x3, k2, 0));           // 2(x2 + x3)
        uint64_t T2 = mad64(x1, k4, T1);                      // 2(2x1 + t1)
        uint64_t T3 = mad64(x3, k4, T0);                      // 2(2x3 + t0)
        uint64_t T4 = shl_add64<2>(T1, T3);                   // 2(4t1 + t3)
        uint64_t T5 = shl_add64<2>(T0, T2);                   // 2(4t0 + t2)
        y0 = add64(T3, T5);
        y1 = T5;
        y2 = add64(T2, T4);
        y3 = T4;
    }

    // V[i] = 2 * (circ(2M4, M4, M4, M4) * s)[i], inputs any u32 (< 2^32): the matrix rows sum to at most 16 * 5 = 80, so
    // every V[i] < 2 * 80 * 2^32 < 2^40.  V never carries a round constant (they are literals of the fused reductions).
    static __device__ __forceinline__ void mds16_2x(uint32_t k2, uint32_t k4, const uint32_t* s, uint64_t* V) {
    #pragma unroll
        for (int g = 0; g < 4; g++)
            mds4_2x(k2, k4, s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3], V[4 * g], V[4 * g + 1], V[4 * g + 2], V[4 * g + 3]);
    #pragma unroll
        for (int j = 0; j < 4; j++) {
            uint64_t sum = add64(add64(V[j], V[j + 4]), add64(V[j + 8], V[j + 12]));
    #pragma unroll
            for (int g = 0; g < 4; g++) V[4 * g + j] = add64(V[4 * g + j], sum);
        }
    }

    // Round constant + canonicalisation in one step.  t = fold2(V) with V the doubled accumulator of a linear layer
    // WITHOUT its round constant: t <= P + HI where HI bounds the accumulator's high word (160 after a full-round
    // layer, < 2^19 after a partial-round one).  With c = P - rc (a compile-time literal):
    //     t >= c :  t - c in [0, P - 1]  (needs c > HI; every constant of this parameter set is < P - 2^19, asserted below)
    //               and t - c + P in [P, 2P - 1]: the minimum is t - c  = t + rc - P
    //     t <  c :  t - c wraps to >= 2^32 - P > 2^31 and t - c + P = t + rc in [0, P - 1]: the minimum is t + rc
    // so min(t - c, t - c + P) is the CANONICAL t + rc mod P in two literal adds (fast class, 2.5 cycles each) and a
    // v_min — against v_mad_u64_u32 (2 * rc folded into V, 5.1) + a literal add + v_min for the unfused form.
    template <uint32_t RC>
    static __device__ __forceinline__ uint32_t canon_rc(uint32_t t) {
        static_assert(RC < P - (1u << 19), "round constant too close to P for the fused reduction");
        constexpr uint32_t c = P - RC;
        uint32_t r = min(t - c, t + (P - c));
        if constexpr (PACE) asm volatile("s_nop 0" : "+v"(r));
        return r;
    }

    template <int R, int I>
    static __device__ __forceinline__ void sbox_full(const uint64_t* V, uint32_t* s) {
        s[I] = pow5(canon_rc<RC_FULL_K[R][I]>(fold2(V[I])));
        if constexpr (I + 1 < 16) sbox_full<R, I + 1>(V, s);
    }
    // the first full round of the second half takes its inputs already folded (from the last partial round)
    template <int I>
    static __device__ __forceinline__ void sbox_full4(uint32_t* s) {
        s[I] = pow5(canon_rc<RC_FULL_K[4][I]>(s[I]));
        if constexpr (I + 1 < 16) sbox_full4<I + 1>(s);
    }

    template <int R>
    static __device__ __forceinline__ void partial_round(uint32_t* s, uint32_t k2, uint32_t k6, const uint32_t* kd) {
        uint32_t u0 = pow5(canon_rc<RC_PARTIAL_K[R]>(s[0]));                 // s[0] <= P + 2^19
        // sum2 = 2 * (u0 + s[1] + ... + s[15]) < 2^37, two chains
        uint64_t a = mul64(u0, k2, 0), b = mul64(s[1], k2, 0);
    #pragma unroll
        for (int i = 2; i < 16; i += 2) { a = mad64(s[i], k2, a); b = mad64(s[i + 1], k2, b); }
        uint64_t sum2 = add64(a, b);
        // 2 * (d_i * s_i + sum), d = (3, 4, 8, ..., 65536): < 2^50, so every fold is <= P + 2^18
        s[0] = fold2(mad64(u0, k6, sum2));
    #pragma unroll
        for (int i = 1; i < 16; i++) s[i] = fold2(mad64(s[i], kd[i], sum2));
    }

    // Everything up to and including the S-box layer of the last full round: s = that layer's outputs (range L2).
    static __device__ __forceinline__ void poseidon2_rounds(uint32_t* s, uint32_t k2, uint32_t k4) {
        uint64_t V[16];
        const uint32_t k6 = opaque(6);
        // s: canonical input.  V never carries a round constant: the constants are literals of the fused reductions.
        mds16_2x(k2, k4, s, V);
        sbox_full<0, 0>(V, s); mds16_2x(k2, k4, s, V);
        sbox_full<1, 0>(V, s); mds16_2x(k2, k4, s, V);
        sbox_full<2, 0>(V, s); mds16_2x(k2, k4, s, V);
        sbox_full<3, 0>(V, s); mds16_2x(k2, k4, s, V);
        // partial rounds: every lane lazily folded (<= P + 2^18), lane 0 goes through the S-box
    #pragma unroll
        for (int i = 0; i < 16; i++) s[i] = fold2(V[i]);
        // 2 * diag: 2^(i+2) for lanes 1..15, as opaque wave-uniform multipliers
        uint32_t kd[16];
        kd[0] = k6;
    #define RSV_KD(i) kd[i] = opaque(4u << (i));
        RSV_KD(1) RSV_KD(2) RSV_KD(3) RSV_KD(4) RSV_KD(5) RSV_KD(6) RSV_KD(7) RSV_KD(8)
        RSV_KD(9) RSV_KD(10) RSV_KD(11) RSV_KD(12) RSV_KD(13) RSV_KD(14) RSV_KD(15)
    #undef RSV_KD
        partial_round<0>(s, k2, k6, kd);  partial_round<1>(s, k2, k6, kd);  partial_round<2>(s, k2, k6, kd);
        partial_round<3>(s, k2, k6, kd);  partial_round<4>(s, k2, k6, kd);  partial_round<5>(s, k2, k6, kd);
        partial_round<6>(s, k2, k6, kd);  partial_round<7>(s, k2, k6, kd);  partial_round<8>(s, k2, k6, kd);
        partial_round<9>(s, k2, k6, kd);  partial_round<10>(s, k2, k6, kd); partial_round<11>(s, k2, k6, kd);
        partial_round<12>(s, k2, k6, kd); partial_round<13>(s, k2, k6, kd);
        sbox_full4<0>(s);      mds16_2x(k2, k4, s, V);
        sbox_full<5, 0>(V, s); mds16_2x(k2, k4, s, V);
        sbox_full<6, 0>(V, s); mds16_2x(k2, k4, s, V);
        sbox_full<7, 0>(V, s);
    }

    static __device__ __forceinline__ void poseidon2_inline(uint32_t* s) {
        const uint32_t k2 = opaque(2), k4 = opaque(4);
        poseidon2_rounds(s, k2, k4);
        uint64_t V[16];
        mds16_2x(k2, k4, s, V);
        // canonical output: fold <= P + 160, so one conditional subtract lands in [0, P); P itself maps to 0
    #pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t t = fold2(V[i]);
            s[i] = min(t, t - P);
        }
    }

    // The last linear layer for ONE half of the state (hi = 0: words 0..7, the rate; 1: words 8..15, the capacity): the four
    // M4 blocks and the column sums are needed either way, the per-word additions, folds and canonicalisations only for
    // the eight words asked for — every hash of the verify pipeline keeps one half of the permutation's output
    // (Poseidon2HalfVar::permute's ignore_left_result / ignore_right_result, primitives/poseidon31/src/lib.rs:251-288).
    template <bool HI_CONST = false, bool HI_VALUE = false>
    static __device__ __forceinline__ void poseidon2_inline_half(uint32_t* s, uint32_t hi, uint32_t* out8) {
        if (HI_CONST) hi = HI_VALUE ? 1u : 0u;
        const uint32_t k2 = opaque(2), k4 = opaque(4);
        poseidon2_rounds(s, k2, k4);
        uint64_t V[16];
    #pragma unroll
        for (int g = 0; g < 4; g++)
            mds4_2x(k2, k4, s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3], V[4 * g], V[4 * g + 1], V[4 * g + 2], V[4 * g + 3]);
        // one column at a time (the column sum lives in two registers), the branch outside the loop: the instance must fit
        // the 40 caller-saved registers v0..v39 like poseidon2(), or the Merkle kernels spill around every call
        if (hi) {  // wave-uniform at every call site
    #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t sum = add64(add64(V[j], V[j + 4]), add64(V[j + 8], V[j + 12]));
                const uint32_t t0 = fold2(add64(V[8 + j], sum)), t1 = fold2(add64(V[12 + j], sum));
                out8[j] = min(t0, t0 - P);
                out8[4 + j] = min(t1, t1 - P);
            }
        } else {
    #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t sum = add64(add64(V[j], V[j + 4]), add64(V[j + 8], V[j + 12]));
                const uint32_t t0 = fold2(add64(V[j], sum)), t1 = fold2(add64(V[4 + j], sum));
                out8[j] = min(t0, t0 - P);
                out8[4 + j] = min(t1, t1 - P);
            }
        }
    }
};
// (tools/perm_lab.hip, k_permute: the paced form)
__device__ __forceinline__ void poseidon2_inline(uint32_t* s) { PermT<true>::poseidon2_inline(s); }

#ifdef RSV_COUNT_PERMS
// Diagnostic build only (make count): executed permutations per kernel tag — [2t] active lanes, [2t+1] wave-level calls.
__device__ unsigned long long g_perm_counter[16];
__shared__ unsigned s_perm_tag;
#define RSV_TAG(k) do { if (threadIdx.x == 0) s_perm_tag = (k); __syncthreads(); } while (0)
#else
#define RSV_TAG(k) do { } while (0)
#endif

// Out-of-line instance with the whole output state: rsv_poseidon2_permute*, the PoseidonFlow kernels.
__device__ __noinline__ State16 poseidon2(State16 st) {
#ifdef RSV_COUNT_PERMS
    {
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) {
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u)], (unsigned long long)__builtin_popcountll(m));
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u) + 1], 1ull);
        }
    }
#endif
    poseidon2_inline(st.s);
    return st;
}

// Poseidon2HalfVar::permute semantics (primitives/poseidon31/src/lib.rs:282-311):
// state = left || right; returns rate = out[0..8] and capacity = out[8..16].
struct Hash8 {
    uint32_t w[8];
};
__device__ __forceinline__ State16 join(const Hash8& l, const Hash8& r) {
    State16 st;
#pragma unroll
    for (int i = 0; i < 8; i++) { st.s[i] = l.w[i]; st.s[8 + i] = r.w[i]; }
    return st;
}
__device__ __forceinline__ Hash8 rate_of(const State16& st) {
    Hash8 h;
#pragma unroll
    for (int i = 0; i < 8; i++) h.w[i] = st.s[i];
    return h;
}
__device__ __forceinline__ Hash8 cap_of(const State16& st) {
    Hash8 h;
#pragma unroll
    for (int i = 0; i < 8; i++) h.w[i] = st.s[8 + i];
    return h;
}
__device__ __forceinline__ Hash8 zero8() {
    Hash8 h;
#pragma unroll
    for (int i = 0; i < 8; i++) h.w[i] = 0;
    return h;
}
// (Until round 4 a second form stood beside this one — a single out-of-line instance with a run-time half selector, 51
// VGPRs, the Merkle kernels spilling 32 B per lane around its calls; same step time, 1 GB more traffic.  Its callers moved
// to the template arguments below, so it no longer compiled: removed.)
// One out-of-line instance per output half and pacing.  The Merkle kernels (several waves per SIMD wherever a launch
// fills the machine) call the PACED ones; the lane-form transcript — one wave per SIMD for 65 536 proofs — the unpaced
// ones (k_transcript: 3.21 -> 2.66 ms).  A choice per call site by launch size (tried: small batches' trees unpaced, one
// proof 1.31 -> 1.22 ms) doubles every call in the Merkle kernels, which then spill 80-112 bytes per lane around them and
// lose 4.7 % at 65 536 proofs: not taken; the choice is the call site's own template argument.
template <bool HI, bool PACE>
__device__ __noinline__ Hash8 poseidon2_half_t(State16 st) {
#ifdef RSV_COUNT_PERMS
    {
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) {
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u)], (unsigned long long)__builtin_popcountll(m));
            atomicAdd(&g_perm_counter[2 * (s_perm_tag & 7u) + 1], 1ull);
        }
    }
#endif
    Hash8 h;
    PermT<PACE>::template poseidon2_inline_half<true, HI>(st.s, 0u, h.w);
    return h;
}
// The row form of the same call (poseidon2_row.hpp): the 16 lanes of a DPP row hold the SAME state and the same result,
// each computes one word of it.  For a launch of a few waves, where a lane-form permutation (one lane, sixteen words) is a
// chain of ~5 000 instructions and the row form one of ~1 250.
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
template <bool HI>
__device__ __noinline__ Hash8 poseidon2_row_half(State16 st);
__device__ __noinline__ State16 poseidon2_row_state(State16 st);  // the whole output state (PoseidonFlow records)
// The row form's round constants for the out-of-line instance, in LDS: [round][lane of the row].  A kernel that calls
// poseidon2_half<FORM_ROW> fills the table first (row_rc_init).  Not global memory: a load inside the callee would make it
// wait for every load its caller has in flight (the memory counter is in order; LDS has a counter of its own) — the tree
// kernels fetch the next level's sibling while a level is hashed.  Not an argument either: eight more values live across
// every call, and the callers spill.
__shared__ uint32_t s_row_rc[8][16];
template <int PACE>
__device__ __forceinline__ void row_rc_init() {
    if constexpr (PACE == 2) {
        if (threadIdx.x < 128) s_row_rc[threadIdx.x >> 4][threadIdx.x & 15u] = RC_FULL[threadIdx.x >> 4][threadIdx.x & 15u];
        __syncthreads();
    }
}
// FORM (the `PACE` argument of every helper and kernel below and in merkle.hpp / k_merkle.hpp): 1 the paced lane form,
// 0 the unpaced lane form, 2 (FORM_ROW) the row form on "virtual lanes" of 16 lanes each.
constexpr int FORM_ROW = 2;
template <int PACE = 1>
__device__ __forceinline__ Hash8 poseidon2_half(State16 st, uint32_t hi) {  // hi is a literal at every call site
    if constexpr (PACE == FORM_ROW) return hi ? poseidon2_row_half<true>(st) : poseidon2_row_half<false>(st);
    else return hi ? poseidon2_half_t<true, PACE != 0>(st) : poseidon2_half_t<false, PACE != 0>(st);
}
// whole output state, by form (the PoseidonFlow kernels: lane form paced, or the row form)
template <int PACE = 1>
__device__ __forceinline__ State16 poseidon2_full(State16 st) {
    if constexpr (PACE == FORM_ROW) return poseidon2_row_state(st);
    else return poseidon2(st);
}
// the lane of a kernel's own indexing: a thread (lane forms) or a DPP row of 16 threads that all compute the same (row form)
template <int PACE>
__device__ __forceinline__ uint32_t vlane() { return PACE == FORM_ROW ? threadIdx.x >> 4 : threadIdx.x; }
template <int PACE = 1>
__device__ __forceinline__ Hash8 perm_rate(const Hash8& l, const Hash8& r) { return poseidon2_half<PACE>(join(l, r), 0u); }
template <int PACE = 1>
__device__ __forceinline__ Hash8 perm_cap(const Hash8& l, const Hash8& r) { return poseidon2_half<PACE>(join(l, r), 1u); }
__device__ __forceinline__ bool hash_eq(const Hash8& a, const Hash8& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.w[i] ^ b.w[i];
    return d == 0;
}

}  // namespace rsv
