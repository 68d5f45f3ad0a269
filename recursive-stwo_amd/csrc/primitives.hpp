// primitives.hpp — batch kernels behind the small C-ABI entry points
// (rsv_poseidon2_permute, rsv_poseidon2_half_permute, rsv_merkle_hash_node,
// rsv_merkle_path_root).  One lane per item; 256-thread workgroups; the grid is
// sized to the item count (>> 256 workgroups for any non-trivial batch).
#pragma once
#include "merkle.hpp"

namespace rsv {

// a3: poseidon2_permute over n states (primitives/poseidon31/src/implementation.rs:108-149).
// Persistent grid-stride form: the grid is sized to the machine (not to n), every lane walks states
// tid, tid + stride, ... and fetches its NEXT 64-byte state (four 16-byte loads) before it permutes the current
// one, so the ~2 us HBM latency sits underneath ~6 us of arithmetic instead of in front of it, and no wave pays
// launch + first-touch latency per single permutation (7.97 -> 8.83 G permutations/s on 2^24 states).  Measured and
// rejected: a tile-per-wave variant with fully coalesced 1 KB global accesses and an LDS transpose (8.59 G/s): the
// kernel is VALU-issue bound, the 16-byte-per-lane accesses at 64-byte stride cost nothing that shows.
// FORM (RSV_OPT_PERM_FORM, an experiment's knob; 0 = what production runs): 0 the out-of-line paced instance the verify
// kernels call, 1 the same inlined (+0.2 % at 24 workgroups per CU, +2 % at 6), 2 inlined without the wait states (-4 %);
// WAVES: the launch bound (the kernel takes 62-66 registers whatever it says).
template <int FORM = 0, int WAVES = 1>
__global__ __launch_bounds__(256, WAVES) void k_permute(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                 size_t n, uint32_t* __restrict__ bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 nx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) nx[k] = in[4 * i + k];
    uint32_t over = 0;
    while (true) {
        State16 st;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = nx[k];
            st.s[4 * k] = v.x; st.s[4 * k + 1] = v.y; st.s[4 * k + 2] = v.z; st.s[4 * k + 3] = v.w;
            over |= (v.x >= P) | (v.y >= P) | (v.z >= P) | (v.w >= P);
        }
        const size_t next = i + stride;
        if (next < n) {
#pragma unroll
            for (int k = 0; k < 4; k++) nx[k] = in[4 * next + k];
        }
        if constexpr (FORM == 0) st = poseidon2(st);  // the single out-of-line instance the verify kernels call
        else if constexpr (FORM == 1) PermT<true>::poseidon2_inline(st.s);
        else PermT<false>::poseidon2_inline(st.s);
#pragma unroll
        for (int k = 0; k < 4; k++) out[4 * i + k] = make_uint4(st.s[4 * k], st.s[4 * k + 1], st.s[4 * k + 2], st.s[4 * k + 3]);
        if (next >= n) break;
        i = next;
    }
    if (over && bad) atomicOr(bad, 1u);
}

// a4: Poseidon2HalfVar::permute (primitives/poseidon31/src/lib.rs:282-423)
__global__ __launch_bounds__(256) void k_half_permute(const uint32_t* __restrict__ left,
                                                      const uint32_t* __restrict__ right,
                                                      const uint8_t* __restrict__ swap,
                                                      uint32_t* __restrict__ rate, uint32_t* __restrict__ cap,
                                                      size_t n, uint32_t* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hash8 l = load_hash(left + 8 * i), r = load_hash(right + 8 * i);
    uint32_t over = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) over |= (l.w[k] >= P) | (r.w[k] >= P);
    if (over) atomicOr(bad, 1u);
    bool sw = swap && swap[i];
    State16 st = poseidon2(sw ? join(r, l) : join(l, r));
    if (rate) store_hash(rate + 8 * i, rate_of(st));
    if (cap) store_hash(cap + 8 * i, cap_of(st));
}

// a5: Poseidon31MerkleHasher::hash_node over n nodes (primitives/merkle/src/lib.rs:9-181)
__global__ __launch_bounds__(256) void k_hash_node(const uint32_t* __restrict__ left,
                                                   const uint32_t* __restrict__ right,
                                                   const uint32_t* __restrict__ cols, uint32_t n_cols,
                                                   uint32_t* __restrict__ out, size_t n,
                                                   uint32_t* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t over = 0;
    const uint32_t* c = cols ? cols + (size_t)n_cols * i : nullptr;
    for (uint32_t k = 0; k < n_cols; k++) over |= c[k] >= P;
    Hash8 h;
    if (left) {
        Hash8 l = load_hash(left + 8 * i), r = load_hash(right + 8 * i);
#pragma unroll
        for (int k = 0; k < 8; k++) over |= (l.w[k] >= P) | (r.w[k] >= P);
        h = hash_node(&l, &r, c, n_cols);
    } else {
        h = hash_node(nullptr, nullptr, c, n_cols);
    }
    if (over) atomicOr(bad, 1u);
    store_hash(out + 8 * i, h);
}

// a9: SinglePathMerkleProofVar::verify (components/recursive/data_structures/src/lib.rs:315-354),
// one lane per authentication path; returns the recomputed root.
__global__ __launch_bounds__(256) void k_path_root(const uint32_t* __restrict__ query,
                                                   const uint32_t* __restrict__ sib,
                                                   const uint32_t* __restrict__ cols,
                                                   const uint32_t* __restrict__ n_cols_at, uint32_t depth,
                                                   uint32_t per_path, uint32_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* c = cols + (size_t)per_path * i;
    uint32_t q = query[i];
    Hash8 cur = hash_node(nullptr, nullptr, c, n_cols_at[depth]);
    c += n_cols_at[depth];
    for (uint32_t lvl = 0; lvl < depth; lvl++) {
        uint32_t h = depth - lvl - 1;
        Hash8 s = load_hash(sib + 8 * ((size_t)depth * i + lvl));
        uint32_t nc = n_cols_at[h];
        bool right = (q >> lvl) & 1;
        Hash8 l = right ? s : cur, r = right ? cur : s;
        cur = hash_node(&l, &r, c, nc);
        c += nc;
    }
    store_hash(out + 8 * i, cur);
}

}  // namespace rsv
