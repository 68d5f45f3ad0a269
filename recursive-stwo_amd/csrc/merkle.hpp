// merkle.hpp — Poseidon31 Merkle hasher and Fiat-Shamir channel, device side.
//
// Values side of Poseidon31MerkleHasherVar (primitives/merkle/src/lib.rs:9-181),
// i.e. stwo's Poseidon31MerkleHasher::hash_node(children, columns), and of
// ChannelVar (primitives/channel/src/lib.rs:24-58).  One lane owns one node /
// one channel; every function is a short sequence of calls to the single
// out-of-line poseidon2() instance.
#pragma once
#include "poseidon2.hpp"

namespace rsv {

// 1 when some word of the hash is not a canonical M31 (layout.hpp: canonicity is checked where words are read)
__device__ __forceinline__ uint32_t hash_over(const Hash8& h) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= h.w[i] >= P;
    return o;
}

// PACE (every helper below): which out-of-line permutation instance it calls (poseidon2.hpp) — the paced one for launches
// that put several waves on a SIMD, the unpaced one for a small batch, where a wave is nearly alone.
// hash_m31_columns_get_capacity (primitives/merkle/src/lib.rs:141-181):
// d = 0; for each zero-padded chunk of 8 words: d = perm(chunk || d)[8..16].
// `cols` may be any address space; words are read with plain 4-byte loads.
template <int PACE = 1>
__device__ inline Hash8 sponge_capacity(const uint32_t* cols, uint32_t n) {
    Hash8 d = zero8();
    for (uint32_t off = 0; off < n; off += 8) {
        Hash8 chunk;
#pragma unroll
        for (int i = 0; i < 8; i++) chunk.w[i] = (off + i < n) ? cols[off + i] : 0u;
        d = perm_cap<PACE>(chunk, d);
    }
    return d;
}
// the same, reporting a non-canonical column word through `over`
template <int PACE = 1>
__device__ inline Hash8 sponge_capacity_chk(const uint32_t* cols, uint32_t n, uint32_t& over) {
    Hash8 d = zero8();
    for (uint32_t off = 0; off < n; off += 8) {
        Hash8 chunk;
#pragma unroll
        for (int i = 0; i < 8; i++) chunk.w[i] = (off + i < n) ? cols[off + i] : 0u;
        over |= hash_over(chunk);
        d = perm_cap<PACE>(chunk, d);
    }
    return d;
}
// The 4-word (one QM31) column of the FRI trees: hash_qm31_columns_get_capacity of
// [v, 0] (components/recursive/data_structures/src/lib.rs:408-419).
template <int PACE = 1>
__device__ inline Hash8 sponge_capacity4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    Hash8 chunk = zero8();
    chunk.w[0] = v0; chunk.w[1] = v1; chunk.w[2] = v2; chunk.w[3] = v3;
    return perm_cap<PACE>(chunk, zero8());
}
// leaf: hash_m31_columns_get_rate (primitives/merkle/src/lib.rs:50-91)
template <int PACE = 1>
__device__ inline Hash8 leaf_from_capacity(const Hash8& d) { return perm_rate<PACE>(zero8(), d); }
// hash_tree (primitives/merkle/src/lib.rs:9-11)
template <int PACE = 1>
__device__ inline Hash8 hash_tree(const Hash8& l, const Hash8& r) { return perm_rate<PACE>(l, r); }
// hash_tree_with_swap (primitives/merkle/src/lib.rs:22-30): ONE call site for both orders — a lane-divergent
// `odd ? hash_tree(b, a) : hash_tree(a, b)` would run the permutation twice per wave with half the lanes masked.
template <int PACE = 1>
__device__ inline Hash8 hash_tree_swap(const Hash8& self, const Hash8& sibling, bool self_is_right) {
    State16 st;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        st.s[i] = self_is_right ? sibling.w[i] : self.w[i];
        st.s[8 + i] = self_is_right ? self.w[i] : sibling.w[i];
    }
    return poseidon2_half<PACE>(st, 0u);
}
// combine_hash_tree_with_column (primitives/merkle/src/lib.rs:43-48)
template <int PACE = 1>
__device__ inline Hash8 combine_with_column(const Hash8& tree, const Hash8& col_cap) { return perm_rate<PACE>(tree, col_cap); }

// stwo Poseidon31MerkleHasher::hash_node
template <int PACE = 1>
__device__ inline Hash8 hash_node(const Hash8* l, const Hash8* r, const uint32_t* cols, uint32_t n_cols) {
    if (!l) return leaf_from_capacity<PACE>(sponge_capacity<PACE>(cols, n_cols));
    Hash8 h = hash_tree<PACE>(*l, *r);
    if (n_cols) h = combine_with_column<PACE>(h, sponge_capacity<PACE>(cols, n_cols));
    return h;
}

// ---- PoseidonFlow records (layout.hpp; include/rsv.h: rsv_hints_out::d_flow) --------------------------------------
// Where a proof's records go.  rec == nullptr: emission off.
struct FlowSink {
    uint4* rec;     // [stride][8] uint4 = 32 words per record, already offset to the proof
    uint8_t* swap;  // [stride], already offset to the proof
};
// The caller's buffers (kernel argument)
struct FlowArgs {
    uint32_t* rec;     // [n][stride][32]   (nullptr: no flow)
    uint8_t* swap;     // [n][stride]
    uint32_t* count;   // [n] or nullptr: records of proof i (0: rejected by the parser, or the stride is too small)
    uint32_t stride;
    __device__ FlowSink sink(uint32_t p) const {
        return FlowSink{reinterpret_cast<uint4*>(rec) + (size_t)p * stride * 8, swap + (size_t)p * stride};
    }
};
// One invocation: the two input halves AS GIVEN (the accelerator applies the swap:
// primitives/poseidon31/src/lib.rs:385-415), the output state, the swap bit.  One whole 128-byte line per record.
__device__ __forceinline__ void flow_put(const FlowSink& f, uint32_t idx, const Hash8& l, const Hash8& r, const State16& out, uint32_t swap) {
    uint4* q = f.rec + (size_t)idx * 8;
    q[0] = make_uint4(l.w[0], l.w[1], l.w[2], l.w[3]);
    q[1] = make_uint4(l.w[4], l.w[5], l.w[6], l.w[7]);
    q[2] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
    q[3] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
#pragma unroll
    for (int k = 0; k < 4; k++) q[4 + k] = make_uint4(out.s[4 * k], out.s[4 * k + 1], out.s[4 * k + 2], out.s[4 * k + 3]);
    f.swap[idx] = (uint8_t)swap;
}
// permute(left, right, swap) with its record: returns the output state.  PACE: the permutation instance (1 the paced lane
// form, FORM_ROW the row form on virtual lanes: the 16 threads of a row then write the same record)
template <int PACE = 1>
__device__ inline State16 flow_perm(const FlowSink& f, uint32_t idx, const Hash8& l, const Hash8& r, bool swap) {
    State16 st;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        st.s[i] = swap ? r.w[i] : l.w[i];
        st.s[8 + i] = swap ? l.w[i] : r.w[i];
    }
    st = poseidon2_full<PACE>(st);
    flow_put(f, idx, l, r, st, swap ? 1u : 0u);
    return st;
}
// hash_m31_columns_get_capacity with records idx, idx + 1, ... (one per 8-word chunk)
template <int PACE = 1>
__device__ inline Hash8 flow_sponge_capacity(const FlowSink& f, uint32_t idx, const uint32_t* cols, uint32_t n) {
    Hash8 d = zero8();
    for (uint32_t off = 0; off < n; off += 8) {
        Hash8 chunk;
#pragma unroll
        for (int i = 0; i < 8; i++) chunk.w[i] = (off + i < n) ? cols[off + i] : 0u;
        d = cap_of(flow_perm<PACE>(f, idx++, chunk, d, false));
    }
    return d;
}
// hash_qm31_columns_get_capacity(&[v, 0]): one record
template <int PACE = 1>
__device__ inline Hash8 flow_capacity4(const FlowSink& f, uint32_t idx, const uint32_t* v) {
    Hash8 chunk = zero8();
    chunk.w[0] = v[0]; chunk.w[1] = v[1]; chunk.w[2] = v[2]; chunk.w[3] = v[3];
    return cap_of(flow_perm<PACE>(f, idx, chunk, zero8(), false));
}

// ChannelVar (primitives/channel/src/lib.rs:24-58).  PACE: the permutation instance its operations call (poseidon2.hpp)
template <int PACE = 1>
struct Channel {
    Hash8 digest;
    uint32_t n_sent;
    FlowSink flow;      // PoseidonFlow records of the channel operations (rec == nullptr: off)
    uint32_t flow_idx;
    __device__ void init() { digest = zero8(); n_sent = 0; flow.rec = nullptr; flow.swap = nullptr; flow_idx = 0; }
    __device__ void mix(const Hash8& left) {
        if (flow.rec) digest = cap_of(flow_perm(flow, flow_idx++, left, digest, false));
        else digest = perm_cap<PACE>(left, digest);
        n_sent = 0;
    }
    __device__ void mix_two(QM31 f, QM31 g) {
        Hash8 l;
        l.w[0] = f.a.a; l.w[1] = f.a.b; l.w[2] = f.b.a; l.w[3] = f.b.b;
        l.w[4] = g.a.a; l.w[5] = g.a.b; l.w[6] = g.b.a; l.w[7] = g.b.b;
        mix(l);
    }
    __device__ void mix_one(QM31 f) { mix_two(f, q_zero()); }
    // returns the 8 rate words = two QM31
    __device__ Hash8 draw() {
        Hash8 l = zero8();
        l.w[0] = n_sent++;
        if (flow.rec) return rate_of(flow_perm(flow, flow_idx++, l, digest, false));
        return perm_rate<PACE>(l, digest);
    }
};
__device__ __forceinline__ QM31 q_lo(const Hash8& h) { return q_mk(h.w[0], h.w[1], h.w[2], h.w[3]); }
__device__ __forceinline__ QM31 q_hi(const Hash8& h) { return q_mk(h.w[4], h.w[5], h.w[6], h.w[7]); }

__device__ __forceinline__ Hash8 load_hash(const uint32_t* p) {
    Hash8 h;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        uint4 a = q[0], b = q[1];
        h.w[0] = a.x; h.w[1] = a.y; h.w[2] = a.z; h.w[3] = a.w;
        h.w[4] = b.x; h.w[5] = b.y; h.w[6] = b.z; h.w[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) h.w[i] = p[i];
    }
    return h;
}
__device__ __forceinline__ Hash8 load_hash_chk(const uint32_t* p, uint32_t& over) {
    Hash8 h = load_hash(p);
    over |= hash_over(h);
    return h;
}
__device__ __forceinline__ void store_hash(uint32_t* p, const Hash8& h) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = h.w[i];
}

}  // namespace rsv
