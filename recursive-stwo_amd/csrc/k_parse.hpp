// k_parse.hpp — wire-format walk (k_parse) and the canonicity fallback scan (k_rescan).  Part of the pipeline described in verify.hpp.
#pragma once
#include "verify_common.hpp"

namespace rsv {

// ------------------------------------------------------------------ k_parse
// Walks the length prefixes of one proof and records where every section lives.
struct WordReader {
    const uint32_t* w;
    uint32_t n, pos;
    bool ok;
    __device__ uint32_t u32() {
        if (!ok || pos >= n) { ok = false; return 0; }
        return w[pos++];
    }
    // u64 that must fit 32 bits
    __device__ uint32_t len() {
        uint32_t lo = u32(), hi = u32();
        if (hi != 0) ok = false;
        return lo;
    }
    __device__ uint32_t skip(uint32_t words) {
        uint32_t at = pos;
        if (!ok || words > n - pos) { ok = false; return at; }
        pos += words;
        return at;
    }
};

__device__ inline void parse_decommit(WordReader& r, uint32_t& off, uint32_t& cnt) {
    cnt = r.len();
    if (cnt > (1u << 20)) r.ok = false;
    off = r.skip(r.ok ? 8u * cnt : 0u);
    if (r.len() != 0) r.ok = false;  // column_witness must be empty (components/hints/src/decommit.rs:71)
}
__device__ inline void parse_fri_layer(WordReader& r, FriLayerRef& l) {
    l.wit_n = r.len();
    if (l.wit_n > (1u << 20)) r.ok = false;
    l.wit_off = r.skip(r.ok ? 4u * l.wit_n : 0u);
    parse_decommit(r, l.hash_off, l.hash_n);
    l.commit_off = r.skip(8);
}

__global__ __launch_bounds__(64) void k_parse(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                              uint32_t n, CfgSet cfg, ProofMeta* __restrict__ metas,
                                              ProofCtx* __restrict__ ctxs, uint32_t* __restrict__ summary,
                                              uint32_t* __restrict__ shape) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ProofMeta& m = metas[p];
    ctxs[p].flags = 0;  // (front_over belongs to the front half of a split transcript, which may be running now)
    shape[2 * p] = 0;
    shape[2 * p + 1] = 0;
    uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    m.reason = R_PARSE;
    m.nq = 0; m.M = 0; m.n_inner = 0;
    if (o1 < o0 || ((o0 | o1) & 3) || (o1 - o0) > (1ull << 30)) return;
    WordReader r{reinterpret_cast<const uint32_t*>(blob + o0), (uint32_t)((o1 - o0) >> 2), 0, true};
    m.n_words = r.n;
    if (r.n < SAMPLES.end + 8) return;
    m.lp = r.w[W_LP]; m.lq = r.w[W_LQ];
    m.pow_bits = r.w[W_POW_BITS]; m.blowup = r.w[W_BLOWUP]; m.log_last = r.w[W_LOG_LAST];
    uint32_t nq = r.w[W_NQ];
    if (r.w[W_NQ + 1] != 0 || nq == 0 || nq > MAXQ) return;
    {
        const uint32_t ci = cfg.cfg_of ? cfg.cfg_of[p] : 0u;
        if (ci >= cfg.n) return;
        // dynamic index into a by-value kernel argument: select with a uniform loop instead of scratch
        uint32_t want[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < (uint32_t)MAX_CFGS; k++)
            if (k == ci) { want[0] = cfg.c[k][0]; want[1] = cfg.c[k][1]; want[2] = cfg.c[k][2]; want[3] = cfg.c[k][3]; }
        if (want[0] != m.pow_bits || want[1] != m.blowup || want[2] != m.log_last || want[3] != nq) return;
    }
    uint32_t b = m.blowup, last = m.log_last;
    if (m.lp < 1 || m.lq < 1 || m.lp > 28 || m.lq > 28 || b < 1 || b > 16 || last > 16 || m.pow_bits > 30) return;
    uint32_t A = m.lp + b, B = m.lq + b, M = umax(m.lp + 1, m.lq + 2) + b;
    if (M > MAX_LOG) return;
    if (A < last + b + 1 || B < last + b + 1) return;
    if (r.w[W_NCOMMIT] != 4 || r.w[W_NCOMMIT + 1] != 0 || r.w[W_NTREES] != 4 || r.w[W_NTREES + 1] != 0) return;
    // constant-shape sampled_values: 4 trees of 50/60/16/8 columns with 1 or 2 mask points
    uint32_t c_all = 0;
    bool ok = true;
    for (int t = 0; t < 4; t++) {
        ok &= r.w[SAMPLES.tree_prefix[t]] == tree_cols(t) && r.w[SAMPLES.tree_prefix[t] + 1] == 0;
        for (uint32_t c = 0; c < tree_cols(t); c++, c_all++)
            ok &= r.w[SAMPLES.col_prefix[c_all]] == n_samples_of(t, (int)c) && r.w[SAMPLES.col_prefix[c_all] + 1] == 0;
    }
    if (!ok) return;
    r.pos = SAMPLES.end;
    if (r.len() != 4) return;
    for (int t = 0; t < 4; t++) parse_decommit(r, m.hw_off[t], m.hw_n[t]);
    if (r.len() != 4) return;
    for (int t = 0; t < 4; t++) {
        m.qv_n[t] = r.len();
        if (m.qv_n[t] > (1u << 22)) r.ok = false;
        m.qv_off[t] = r.skip(r.ok ? m.qv_n[t] : 0u);
    }
    m.nonce_off = r.skip(2);
    parse_fri_layer(r, m.first);
    uint32_t n_inner = r.len();
    if (!r.ok || n_inner != M - 1 - (last + b) || n_inner > MAX_INNER) return;
    for (uint32_t i = 0; i < n_inner; i++) parse_fri_layer(r, m.inner[i]);
    m.last_n = r.len();
    if (!r.ok || m.last_n != (1u << last)) return;  // components/hints/src/fiat_shamir.rs:195-198
    m.last_off = r.skip(4u * m.last_n);
    (void)r.u32();  // last_layer_poly.log_size
    if (!r.ok || r.pos != r.n) return;
    m.nq = nq; m.n_inner = n_inner; m.A = A; m.B = B; m.M = M;
    m.reason = R_OK;
    atomicMax(&summary[0], nq);
    atomicMax(&summary[1], M);
    atomicMax(&summary[2], n_inner);
    atomicMax(&summary[3], 64u - (last + b + 1u));  // 64 - (lowest data / leaf level of any tree)
    // shape word for host-side bucketing + "is the batch uniform" summary (max of x and of ~x)
    const uint32_t sw = nq | (M << 8) | (n_inner << 16) | ((last + b + 1u) << 24);
    // second word: the column log sizes (two proofs with equal first words can still differ in A / B, and lanes of
    // one wavefront should walk trees of ONE geometry: the host orders the slots of a bucket by both words)
    const uint32_t sw2 = A | (B << 8);
    shape[2 * p] = sw;
    shape[2 * p + 1] = sw2;
    atomicMax(&summary[4], sw);
    atomicMax(&summary[5], ~sw);
    atomicMax(&summary[6], sw2);
    atomicMax(&summary[7], ~sw2);
}

// ----------------------------------------------------------------- k_rescan
// Canonicity fallback (layout.hpp, F_RESCAN): a proof in which some stage found a witness list of the wrong length —
// already rejected by that stage — is read once in full here, so that a non-canonical word in the part of the list
// nobody consumed still gives RSV_R_PARSE, the reason include/rsv.h defines for it.  Every field element of a proof must be a
// canonical M31 word (< P); exempt are the words that are not field elements and that nothing else constrains: the
// two halves of the proof-of-work nonce, and the proof's final word, last_layer_poly.log_size — the reference never
// reads it (it takes the size from coeffs.len(), components/hints/src/folding.rs:573, fiat_shamir.rs:196-200), so any
// u32 there verifies.  One lane looks at one proof's flag; the wave then reads the flagged proofs of its 64 one after
// the other with 16-byte coalesced loads.  In a batch of well-formed and bit-flipped proofs nothing is flagged and the
// kernel reads 8 bytes per proof (rounds 1-2 read every proof a second time: 7.7 GB per 65 536-proof step).
// force: every parsed proof is read (the single-proof probe rsv_transcript, which runs no Merkle stage).
__global__ __launch_bounds__(256) void k_rescan(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                uint32_t n, const ProofMeta* __restrict__ metas,
                                                ProofCtx* __restrict__ ctxs, uint32_t force) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t mine = blockIdx.x * blockDim.x + threadIdx.x;
    const bool want = mine < n && metas[mine].reason == R_OK && (force || (ctxs[mine].flags & F_RESCAN));
    unsigned long long todo = __ballot(want);
    while (todo) {
        const uint32_t src = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1ull;
        const uint32_t p = mine - lane + src;
        const ProofMeta& m = metas[p];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
        const uint32_t nw = m.n_words, nonce = m.nonce_off, last = nw - 1;  // k_parse ends exactly on the final word
        auto exempt = [&](uint32_t i) { return i == nonce || i == nonce + 1 || i == last; };
        uint32_t bad = 0;
        // align the vector loop to 16 bytes
        uint32_t head = (uint32_t)(((16 - (reinterpret_cast<uintptr_t>(w) & 15)) & 15) >> 2);
        head = umin(head, nw);
        if (lane < head) bad |= (w[lane] >= P) && !exempt(lane);
        const uint4* v = reinterpret_cast<const uint4*>(w + head);
        uint32_t nv = (nw - head) >> 2;
        for (uint32_t i = lane; i < nv; i += 64) {
            uint4 x = v[i];
            uint32_t base = head + 4 * i;
            uint32_t o = (x.x >= P) | ((x.y >= P) << 1) | ((x.z >= P) << 2) | ((x.w >= P) << 3);
            if (o) {
                for (int k = 0; k < 4; k++)
                    if (((o >> k) & 1) && !exempt(base + k)) bad = 1;
            }
        }
        uint32_t tail = head + 4 * nv;
        if (tail + lane < nw) bad |= (w[tail + lane] >= P) && !exempt(tail + lane);
        if (__any(bad) && lane == 0) atomicOr(&ctxs[p].flags, 1u << R_PARSE);
    }
}

}  // namespace rsv
