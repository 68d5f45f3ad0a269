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
// cross-lane traffic; round constants are wave-uniform and are fetched with
// scalar loads (s_load_dwordx16 from constant memory -> SGPR operands), the
// partial-round diagonal is a 31-bit rotate.  The rounds are real loops
// (not unrolled) so one instance of the permutation is ~14 KB of code and stays
// resident in the instruction cache.
#pragma once
#include "field.hpp"

namespace rsv {

struct State16 {
    uint32_t s[16];
};

__constant__ uint32_t RC_FULL[8][16] = {
    {0x768bab52, 0x70e0ab7d, 0x3d266c8a, 0x6da42045, 0x600fef22, 0x41dace6b, 0x64f9bdd4, 0x5d42d4fe,
     0x76b1516d, 0x6fc9a717, 0x70ac4fb6, 0x00194ef6, 0x22b644e2, 0x1f7916d5, 0x47581be2, 0x2710a123},
    {0x6284e867, 0x018d3afe, 0x5df99ef3, 0x4c1e467b, 0x566f6abc, 0x2994e427, 0x538a6d42, 0x5d7bf2cf,
     0x7fda2dab, 0x0fd854c4, 0x46922fca, 0x3d7763a1, 0x19fd05ca, 0x0a4bbb43, 0x15075851, 0x3d903d76},
    {0x2d290ff7, 0x40809fa0, 0x59dac6ec, 0x127927a2, 0x6bbf0ea0, 0x0294140f, 0x24742976, 0x6e84c081,
     0x22484f4a, 0x354cae59, 0x0453ffe1, 0x3f47a3cc, 0x0088204e, 0x6066e109, 0x3b7c4b80, 0x6b55665d},
    {0x3bc4b897, 0x735bf378, 0x508daf42, 0x1884fc2b, 0x7214f24c, 0x7498be0a, 0x1a60e640, 0x3303f928,
     0x29b46376, 0x5c96bb68, 0x65d097a5, 0x1d358e9f, 0x4a9a9017, 0x4724cf76, 0x347af70f, 0x1e77e59a},
    {0x57090613, 0x1fa42108, 0x17bbef50, 0x1ff7e11c, 0x047b24ca, 0x4e140275, 0x4fa086f5, 0x079b309c,
     0x1159bd47, 0x6d37e4e5, 0x075d8dce, 0x12121ca0, 0x7f6a7c40, 0x68e182ba, 0x5493201b, 0x0444a80e},
    {0x0064f4c6, 0x6467abe6, 0x66975762, 0x2af68f9b, 0x345b33be, 0x1b70d47f, 0x053db717, 0x381189cb,
     0x43b915f8, 0x20df3694, 0x0f459d26, 0x77a0e97b, 0x2f73e739, 0x1876c2f9, 0x65a0e29a, 0x4cabefbe},
    {0x5abd1268, 0x4d34a760, 0x12771799, 0x69a0c9ac, 0x39091e55, 0x7f611cd0, 0x3af055da, 0x7ac0bbdf,
     0x6e0f3a24, 0x41e3b6f7, 0x49b3756d, 0x568bc538, 0x20c079d8, 0x1701c72c, 0x7670dc6c, 0x5a439035},
    {0x7c93e00e, 0x561fbb4d, 0x1178907b, 0x02737406, 0x32fb24f1, 0x6323b60a, 0x6ab12418, 0x42c99cea,
     0x155a0b97, 0x53d1c6aa, 0x2bd20347, 0x279b3d73, 0x4f5f3c70, 0x0245af6c, 0x238359d3, 0x49966a59}};

__constant__ uint32_t RC_PARTIAL[14] = {0x7f7ec4bf, 0x0421926f, 0x5198e669, 0x34db3148, 0x4368bafd,
                                        0x66685c7f, 0x78d3249a, 0x60187881, 0x76dad67a, 0x0690b437,
                                        0x1ea95311, 0x40e5369a, 0x38f103fc, 0x1d226a21};

__device__ __forceinline__ uint32_t pow5(uint32_t x) {
    uint32_t x2 = m_sqr(x);
    return m_mul(m_sqr(x2), x);
}

// M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] in 8 additions, 2 doublings, 2 x4.
__device__ __forceinline__ void mds4(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    uint32_t t0 = m_add(x0, x1), t1 = m_add(x2, x3);
    uint32_t t2 = m_add(m_dbl(x1), t1), t3 = m_add(m_dbl(x3), t0);
    uint32_t t4 = m_add(m_shl(t1, 2), t3), t5 = m_add(m_shl(t0, 2), t2);
    x0 = m_add(t3, t5);
    x1 = t5;
    x2 = m_add(t2, t4);
    x3 = t4;
}

__device__ __forceinline__ void mds16(uint32_t* s) {
#pragma unroll
    for (int g = 0; g < 4; g++) mds4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t sum = m_add(m_add(s[j], s[j + 4]), m_add(s[j + 8], s[j + 12]));
#pragma unroll
        for (int g = 0; g < 4; g++) s[4 * g + j] = m_add(s[4 * g + j], sum);
    }
}

__device__ __forceinline__ void poseidon2_inline(uint32_t* s) {
    mds16(s);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = pow5(m_add(s[i], RC_FULL[r][i]));
        mds16(s);
    }
#pragma unroll 1
    for (int r = 0; r < 14; r++) {
        s[0] = pow5(m_add(s[0], RC_PARTIAL[r]));
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
        for (int i = 0; i < 16; i++) s[i] = pow5(m_add(s[i], RC_FULL[r][i]));
        mds16(s);
    }
}

// Out-of-line instance shared by every call site of the large kernels.
__device__ __noinline__ State16 poseidon2(State16 st) {
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
__device__ __forceinline__ Hash8 perm_rate(const Hash8& l, const Hash8& r) { return rate_of(poseidon2(join(l, r))); }
__device__ __forceinline__ Hash8 perm_cap(const Hash8& l, const Hash8& r) { return cap_of(poseidon2(join(l, r))); }
__device__ __forceinline__ bool hash_eq(const Hash8& a, const Hash8& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.w[i] ^ b.w[i];
    return d == 0;
}

}  // namespace rsv
