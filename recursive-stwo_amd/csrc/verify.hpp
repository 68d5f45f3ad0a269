// verify.hpp — the batch verification pipeline for Poseidon31-channel
// Plonk-with-Poseidon Circle-STARK proofs, as HIP kernels for gfx950.
//
// Stage            kernel            parallelism                reference
// ---------------  ----------------  -------------------------  ----------------------------------------
// wire format      k_parse           1 lane / proof             bincode of PlonkWithPoseidonProof
// canonicity       k_scan            1 wave / proof, 16 B/lane  (M31 words must be < P)
// transcript       k_transcript      1 lane / proof             components/recursive/fiat_shamir/src/lib.rs:31-176
// OODS identity    k_oods            1 lane / proof             components/recursive/composition/src/**
// decommit plan    k_plan            1 lane / proof             components/hints/src/decommit.rs:53-142, folding.rs:107-212
// quotients+folds  k_query           1 lane / (proof, query)    components/recursive/answer/src/**, folding/src/lib.rs:57-204
// trace trees      k_trace_merkle    1 lane / (proof,tree,query) components/recursive/data_structures/src/lib.rs:315-354
// FRI trees        k_pair_merkle     1 lane / (proof,layer,query) components/recursive/data_structures/src/lib.rs:400-464
// verdict          k_finalize        1 lane / proof             accept bit + first failing stage
//
// The Merkle kernels follow the reference's per-query form: every lane walks
// one authentication path from its leaf to the root, one Poseidon2 permutation
// per level.  Sibling hashes that the reference's host code re-derives from
// other query paths (SinglePathMerkleProof::from_stwo_proof) are exchanged
// between the lanes of a proof through LDS; the remaining ones are gathered
// straight from the proof's hash_witness at the index given by the plan.
#pragma once
#include "circle.hpp"
#include "layout.hpp"
#include "merkle.hpp"
#include "poseidon2_row.hpp"

namespace rsv {

__constant__ SampleTable SAMPLES = make_sample_table();

enum : uint32_t {
    R_OK = 0, R_PARSE = 1, R_POW = 2, R_LOGUP = 3, R_COMPOSITION = 4, R_DUP_QUERY = 5, R_MERKLE_T0 = 6,
    R_FRI_FIRST = 10, R_FRI_INNER = 11, R_FRI_LAST = 12
};

struct PubInput {
    uint32_t idx;
    uint32_t value[4];
};

struct CfgOpt {
    uint32_t present, pow_bits, blowup, log_last, nq;
};

__device__ __forceinline__ QM31 ldq(const uint32_t* p) { return q_mk(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void stq(uint32_t* p, QM31 v) { p[0] = v.a.a; p[1] = v.a.b; p[2] = v.b.a; p[3] = v.b.b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

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
                                              uint32_t n, CfgOpt cfg, ProofMeta* __restrict__ metas,
                                              ProofCtx* __restrict__ ctxs, uint32_t* __restrict__ summary,
                                              uint32_t* __restrict__ shape) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ProofMeta& m = metas[p];
    ctxs[p].flags = 0;
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
    if (cfg.present && (cfg.pow_bits != m.pow_bits || cfg.blowup != m.blowup || cfg.log_last != m.log_last || cfg.nq != nq))
        return;
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

// ------------------------------------------------------------------- k_scan
// Every field element of a proof must be a canonical M31 word (< P); the only
// words exempt are the two halves of the proof-of-work nonce.  One wave per
// proof reads the proof once with 16-byte coalesced loads — this pass is the
// "proof bytes read once" leg of the HBM roofline.
__global__ __launch_bounds__(256) void k_scan(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                              uint32_t n, ProofMeta* __restrict__ metas) {
    uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n) return;
    ProofMeta& m = metas[wave];
    if (m.reason != R_OK) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[wave]);
    uint32_t nw = m.n_words, nonce = m.nonce_off;
    uint32_t bad = 0;
    // align the vector loop to 16 bytes
    uint32_t head = (uint32_t)(((16 - (reinterpret_cast<uintptr_t>(w) & 15)) & 15) >> 2);
    head = umin(head, nw);
    if (lane < head) bad |= (w[lane] >= P) && lane != nonce && lane != nonce + 1;
    const uint4* v = reinterpret_cast<const uint4*>(w + head);
    uint32_t nv = (nw - head) >> 2;
    for (uint32_t i = lane; i < nv; i += 64) {
        uint4 x = v[i];
        uint32_t base = head + 4 * i;
        uint32_t o = (x.x >= P) | ((x.y >= P) << 1) | ((x.z >= P) << 2) | ((x.w >= P) << 3);
        if (o) {
            for (int k = 0; k < 4; k++)
                if (((o >> k) & 1) && base + k != nonce && base + k != nonce + 1) bad = 1;
        }
    }
    uint32_t tail = head + 4 * nv;
    if (tail + lane < nw) bad |= (w[tail + lane] >= P) && (tail + lane) != nonce && (tail + lane) != nonce + 1;
    if (__any(bad) && lane == 0) m.reason = R_PARSE;
}

// ------------------------------------------------------------- k_transcript
// FiatShamirResults::compute (components/recursive/fiat_shamir/src/lib.rs:44-130):
// a strictly sequential chain of channel permutations per proof.
__global__ __launch_bounds__(64) void k_transcript(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                   uint32_t n, const ProofMeta* __restrict__ metas,
                                                   ProofCtx* __restrict__ ctxs) {
    RSV_TAG(1);
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    Channel ch;
    ch.init();
    Hash8 d;
    ch.mix(load_hash(w + W_COMMIT0));
    ch.mix_one(q_from_m(m.lp));  // statement 0: data_structures/src/lib.rs:52-55
    ch.mix_one(q_from_m(m.lq));
    ch.mix(load_hash(w + W_COMMIT0 + 8));
    d = ch.draw();  // lookup elements z, alpha: data_structures/src/lib.rs:242-245
    stq(c.z, q_lo(d)); stq(c.alpha, q_hi(d));
    ch.mix_two(ldq(w + W_PLONK_SUM), ldq(w + W_POSEIDON_SUM));  // statement 1: data_structures/src/lib.rs:85-87
    ch.mix(load_hash(w + W_COMMIT0 + 16));
    d = ch.draw();
    stq(c.rc, q_lo(d));
    ch.mix(load_hash(w + W_COMMIT0 + 24));
    d = ch.draw();
    QM31 t = q_lo(d);
    stq(c.oods_t, t);
    {  // CirclePointQM31Var::from_t (primitives/circle/src/lib.rs:204-219)
        QM31 t2 = q_mul(t, t);
        QM31 inv = q_inv(q_add(t2, q_one()));
        stq(c.oods_x, q_mul(q_sub(q_one(), t2), inv));
        stq(c.oods_y, q_mul(q_dbl(t), inv));
    }
#pragma unroll 1
    for (int k = 0; k < N_SAMPLES; k += 2)  // fiat_shamir/src/lib.rs:68-75
        ch.mix_two(ldq(w + SAMPLES.off[k]), ldq(w + SAMPLES.off[k + 1]));
    d = ch.draw();
    stq(c.after, q_lo(d));
    ch.mix(load_hash(w + m.first.commit_off));
    d = ch.draw();
    stq(c.fri_alpha[0], q_lo(d));
#pragma unroll 1
    for (uint32_t i = 0; i < m.n_inner; i++) {
        ch.mix(load_hash(w + m.inner[i].commit_off));
        d = ch.draw();
        stq(c.fri_alpha[i + 1], q_lo(d));
    }
#pragma unroll 1
    for (uint32_t i = 0; i < m.last_n; i += 2) {  // fiat_shamir/src/lib.rs:94-100
        const uint32_t* cf = w + m.last_off + 4 * i;
        if (i + 1 < m.last_n) ch.mix_two(ldq(cf), ldq(cf + 4));
        else ch.mix_one(ldq(cf));
    }
    // nonce split 22/21/21: data_structures/src/lib.rs:197-213, fiat_shamir/src/lib.rs:102-113
    uint64_t nonce = (uint64_t)w[m.nonce_off] | ((uint64_t)w[m.nonce_off + 1] << 32);
    ch.mix_one(q_mk((uint32_t)(nonce & ((1u << 22) - 1)), (uint32_t)((nonce >> 22) & ((1u << 21) - 1)),
                    (uint32_t)((nonce >> 43) & ((1u << 21) - 1)), 0));
    store_hash(c.pow_digest, ch.digest);
    uint32_t flags = 0;
    if (ch.digest.w[0] & ((1u << m.pow_bits) - 1u)) flags |= 1u << R_POW;  // fiat_shamir/src/lib.rs:115-117
    uint32_t got = 0;  // fiat_shamir/src/lib.rs:119-130
#pragma unroll 1
    while (got < m.nq) {
        d = ch.draw();
        for (int k = 0; k < 8 && got < m.nq; k++) c.raw_q[got++] = d.w[k];
    }
    c.flags = flags;
}

// --------------------------------------------------------- k_transcript_row
// The same transcript with ONE PROOF PER 16-LANE ROW (poseidon2_row.hpp): lane i holds state word i, lanes
// 0..7 are the rate half (what is mixed in / drawn), lanes 8..15 the capacity half, i.e. the channel digest
// (primitives/channel/src/lib.rs:30-58).  A permutation is ~4x shorter in latency and ~4x dearer in issue
// slots than in k_transcript, so the host uses this kernel for small batches, where the 233-step chain —
// not throughput — is the cost.
__global__ __launch_bounds__(256) void k_transcript_row(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                        uint32_t n, const ProofMeta* __restrict__ metas,
                                                        ProofCtx* __restrict__ ctxs) {
    const uint32_t i = threadIdx.x & 15u;
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= n) return;  // whole rows leave together (DPP needs every lane of a live row)
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    const bool rate = i < 8;
    uint32_t dg = 0, n_sent = 0;
    // mix: digest = perm(left || digest)[8..16]
    auto mix = [&](uint32_t left_word) {
        uint32_t out = poseidon2_row(rate ? left_word : dg, i);
        if (!rate) dg = out;
        n_sent = 0;
    };
    // draw: perm([n_sent, 0 x 7] || digest)[0..8]; the digest is not advanced
    auto draw = [&]() {
        uint32_t out = poseidon2_row(rate ? (i == 0 ? n_sent : 0u) : dg, i);
        n_sent++;
        return out;
    };
    auto mix_words = [&](const uint32_t* src, uint32_t n_words) { mix((rate && i < n_words) ? src[i] : 0u); };
    auto store_felt = [&](uint32_t* dst, uint32_t out) { if (i < 4) dst[i] = out; };
    uint32_t out;
    mix_words(w + W_COMMIT0, 8);
    mix(i == 0 ? m.lp : 0u);  // statement 0: data_structures/src/lib.rs:52-55
    mix(i == 0 ? m.lq : 0u);
    mix_words(w + W_COMMIT0 + 8, 8);
    out = draw();  // lookup elements z, alpha
    if (i < 4) c.z[i] = out; else if (i < 8) c.alpha[i - 4] = out;
    mix_words(w + W_PLONK_SUM, 8);  // statement 1: the two total sums are adjacent words 2..10
    mix_words(w + W_COMMIT0 + 16, 8);
    out = draw();
    store_felt(c.rc, out);
    mix_words(w + W_COMMIT0 + 24, 8);
    out = draw();
    store_felt(c.oods_t, out);
    {  // CirclePointQM31Var::from_t (primitives/circle/src/lib.rs:204-219); every lane computes it
        const int base = (int)((threadIdx.x & 63u) & ~15u);
        QM31 t = q_mk(__shfl(out, base + 0), __shfl(out, base + 1), __shfl(out, base + 2), __shfl(out, base + 3));
        QM31 t2 = q_mul(t, t);
        QM31 inv = q_inv(q_add(t2, q_one()));
        if (i == 0) {
            stq(c.oods_x, q_mul(q_sub(q_one(), t2), inv));
            stq(c.oods_y, q_mul(q_dbl(t), inv));
        }
    }
#pragma unroll 1
    for (int k = 0; k < N_SAMPLES; k += 2)  // fiat_shamir/src/lib.rs:68-75
        mix(rate ? w[SAMPLES.off[k + (i >> 2)] + (i & 3u)] : 0u);
    out = draw();
    store_felt(c.after, out);
    mix_words(w + m.first.commit_off, 8);
    out = draw();
    store_felt(c.fri_alpha[0], out);
#pragma unroll 1
    for (uint32_t l = 0; l < m.n_inner; l++) {
        mix_words(w + m.inner[l].commit_off, 8);
        out = draw();
        store_felt(c.fri_alpha[l + 1], out);
    }
#pragma unroll 1
    for (uint32_t k = 0; k < m.last_n; k += 2) {  // fiat_shamir/src/lib.rs:94-100 (odd tail: second felt = 0)
        const uint32_t left = 4 * (m.last_n - k);
        mix_words(w + m.last_off + 4 * k, left < 8 ? left : 8u);
    }
    // nonce split 22/21/21: data_structures/src/lib.rs:197-213, fiat_shamir/src/lib.rs:102-113
    const uint64_t nonce = (uint64_t)w[m.nonce_off] | ((uint64_t)w[m.nonce_off + 1] << 32);
    mix(i == 0 ? (uint32_t)(nonce & ((1u << 22) - 1)) : i == 1 ? (uint32_t)((nonce >> 22) & ((1u << 21) - 1))
        : i == 2 ? (uint32_t)((nonce >> 43) & ((1u << 21) - 1)) : 0u);
    if (!rate) c.pow_digest[i - 8] = dg;
    if (i == 8) c.flags = (dg & ((1u << m.pow_bits) - 1u)) ? (1u << R_POW) : 0u;  // fiat_shamir/src/lib.rs:115-117
#pragma unroll 1
    for (uint32_t got = 0; got < m.nq; got += 8) {  // fiat_shamir/src/lib.rs:119-130
        out = draw();
        if (rate && got + i < m.nq) c.raw_q[got + i] = out;
    }
}

// ------------------------------------------------------------------- k_oods
// Logup total-sum check (fiat_shamir/src/lib.rs:133-141) and the OODS
// composition identity (components/recursive/composition/src/**).
struct EvalCtx {
    QM31 rc, acc, dinv, z, alpha, alpha2, shift;
    QM31 fp[5], fq[5];
    int n_fracs;
    const uint32_t* w;
    int inter;  // next interaction sample index
    __device__ QM31 smp(int k) const { return ldq(w + SAMPLES.off[k]); }
    // data_structures.rs:26-28,166-169
    __device__ void constraint(QM31 v) { acc = q_add(q_mul(acc, rc), q_mul(v, dinv)); }
    // data_structures.rs:147-164
    __device__ void relation(QM31 mult, QM31 v0, QM31 v1) {
        fp[n_fracs] = mult;
        fq[n_fracs++] = q_sub(q_add(v0, q_mul(alpha, v1)), z);
    }
    __device__ void relation(QM31 mult, QM31 v0, QM31 v1, QM31 v2) {
        fp[n_fracs] = mult;
        fq[n_fracs++] = q_sub(q_add(q_add(v0, q_mul(alpha, v1)), q_mul(alpha2, v2)), z);
    }
    // data_structures.rs:171-210
    __device__ void finalize_logup(int batch) {
        int n_batches = (n_fracs + batch - 1) / batch;
        QM31 prev = q_zero();
        for (int bi = 0; bi < n_batches; bi++) {
            int lo = bi * batch, hi = lo + batch < n_fracs ? lo + batch : n_fracs;
            QM31 pp = fp[lo], qq = fq[lo];
            for (int k = lo + 1; k < hi; k++) {
                pp = q_add(q_mul(pp, fq[k]), q_mul(fp[k], qq));
                qq = q_mul(qq, fq[k]);
            }
            if (bi < n_batches - 1) {
                QM31 cur = q_combine_ef(smp(inter), smp(inter + 1), smp(inter + 2), smp(inter + 3));
                inter += 4;
                constraint(q_sub(q_mul(q_sub(cur, prev), qq), pp));
                prev = cur;
            } else {
                QM31 prev_row = q_combine_ef(smp(inter), smp(inter + 2), smp(inter + 4), smp(inter + 6));
                QM31 cur = q_combine_ef(smp(inter + 1), smp(inter + 3), smp(inter + 5), smp(inter + 7));
                inter += 8;
                QM31 diff = q_sub(q_sub(cur, prev_row), prev);
                constraint(q_sub(q_mul(q_add(diff, shift), qq), pp));
            }
        }
    }
};

__device__ inline QM31 q_double_x(QM31 x, uint32_t times) {
#pragma unroll 1
    for (uint32_t i = 0; i < times; i++) x = q_sub(q_dbl(q_mul(x, x)), q_one());
    return x;
}
__device__ __forceinline__ QM31 q_pow5(QM31 x) {
    QM31 x2 = q_mul(x, x);
    return q_mul(q_mul(x2, x2), x);
}
// poseidon.rs:12-71 over QM31
__device__ inline void q_m4(QM31* x) {
    QM31 t0 = q_add(x[0], x[1]), t02 = q_dbl(t0), t1 = q_add(x[2], x[3]), t12 = q_dbl(t1);
    QM31 t2 = q_add(q_dbl(x[1]), t1), t3 = q_add(q_dbl(x[3]), t0);
    QM31 t4 = q_add(q_dbl(t12), t3), t5 = q_add(q_dbl(t02), t2);
    x[0] = q_add(t3, t5); x[1] = t5; x[2] = q_add(t2, t4); x[3] = t4;
}
__device__ __noinline__ void q_external(QM31* s) {
    for (int g = 0; g < 4; g++) q_m4(s + 4 * g);
    for (int j = 0; j < 4; j++) {
        QM31 sum = q_add(q_add(s[j], s[j + 4]), q_add(s[j + 8], s[j + 12]));
        for (int g = 0; g < 4; g++) s[4 * g + j] = q_add(s[4 * g + j], sum);
    }
}
__device__ __noinline__ void q_internal(QM31* s) {
    QM31 sum = s[0];
    for (int i = 1; i < 16; i++) sum = q_add(sum, s[i]);
    s[0] = q_add(s[0], q_add(q_dbl(s[0]), sum));
    for (int i = 1; i < 16; i++) s[i] = q_add(q_mul_m(s[i], 1u << (i + 1)), sum);
}

__global__ __launch_bounds__(64) void k_oods(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                             uint32_t n, const ProofMeta* __restrict__ metas,
                                             ProofCtx* __restrict__ ctxs, const PubInput* __restrict__ pi,
                                             uint32_t n_pi) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    uint32_t flags = 0;
    QM31 z = ldq(c.z), alpha = ldq(c.alpha), plonk_sum = ldq(w + W_PLONK_SUM), poseidon_sum = ldq(w + W_POSEIDON_SUM);
    {  // fiat_shamir/src/lib.rs:133-141
        QM31 sum = q_zero();
        for (uint32_t i = 0; i < n_pi; i++) {
            QM31 dnm = q_sub(q_add(ldq(pi[i].value), q_mul_m(alpha, pi[i].idx % P)), z);
            sum = q_add(sum, q_inv(dnm));
        }
        if (!q_eq(q_add(q_add(sum, poseidon_sum), plonk_sum), q_zero())) flags |= 1u << R_LOGUP;
    }
    EvalCtx e;
    e.rc = ldq(c.rc); e.acc = q_zero(); e.z = z; e.alpha = alpha; e.alpha2 = q_mul(alpha, alpha); e.w = w;
    QM31 ox = ldq(c.oods_x);
    const QM31 one = q_one();
    {  // plonk.rs:8-82 — preprocessed samples 0..10, trace samples 50..62, interaction samples 110..122
        e.dinv = q_inv(q_double_x(ox, m.lp - 1));  // coset_vanishing: composition/src/lib.rs:18-29
        e.shift = q_mul_m(plonk_sum, m_inv(1u << m.lp));  // data_structures.rs:67-68
        e.inter = S_T2; e.n_fracs = 0;
        const int pre = S_T0, tr = S_T1;
        QM31 enforce = e.smp(pre + 9), op = e.smp(pre + 3);
        e.constraint(q_mul(enforce, e.smp(tr + 9)));
        e.constraint(q_mul(enforce, e.smp(tr + 10)));
        e.constraint(q_mul(enforce, e.smp(tr + 11)));
        QM31 a = q_combine_ef(e.smp(tr + 0), e.smp(tr + 1), e.smp(tr + 2), e.smp(tr + 3));
        QM31 b = q_combine_ef(e.smp(tr + 4), e.smp(tr + 5), e.smp(tr + 6), e.smp(tr + 7));
        QM31 cc = q_combine_ef(e.smp(tr + 8), e.smp(tr + 9), e.smp(tr + 10), e.smp(tr + 11));
        e.constraint(q_sub(q_sub(cc, q_mul(op, q_add(a, b))), q_mul(q_mul(q_sub(one, op), a), b)));
        e.relation(e.smp(pre + 4), a, e.smp(pre + 0));
        e.relation(e.smp(pre + 5), b, e.smp(pre + 1));
        e.relation(e.smp(pre + 6), cc, e.smp(pre + 2));
        e.relation(q_neg(e.smp(pre + 8)), e.smp(pre + 7), a, b);
        e.finalize_logup(2);
    }
    {  // poseidon.rs:73-241 — preprocessed 10..50, trace 62..110, interaction samples 122..134
        e.dinv = q_inv(q_double_x(ox, m.lq - 1));
        e.shift = q_mul_m(poseidon_sum, m_inv(1u << m.lq));
        e.inter = S_T2 + 12; e.n_fracs = 0;
        const int pre = S_T0 + 10, in = S_T1 + 12, mid = in + 16, out = in + 32;
        const int rc0 = pre + 4, rc1 = pre + 20;
        QM31 is_first = e.smp(pre), is_last = e.smp(pre + 1), is_full = e.smp(pre + 2), round_id = e.smp(pre + 3);
        QM31 not_first = q_sub(one, is_first), not_last = q_sub(one, is_last), is_partial = q_sub(not_first, is_full);
        QM31 swap_val = e.smp(mid), one_minus_swap = q_sub(one, swap_val);
        QM31 st[16];
        for (int i = 0; i < 16; i++) {
            QM31 lo = e.smp(in + (i & 7)), hi = e.smp(in + (i & 7) + 8);
            st[i] = i < 8 ? q_add(q_mul(lo, one_minus_swap), q_mul(hi, swap_val))
                          : q_add(q_mul(lo, swap_val), q_mul(hi, one_minus_swap));
        }
        q_external(st);
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_first, q_sub(st[i], e.smp(out + i))));
        for (int i = 0; i < 16; i++) {
            QM31 full = q_pow5(q_add(e.smp(in + i), e.smp(rc0 + i)));
            QM31 mi = e.smp(mid + i);
            e.constraint(q_mul(is_full, q_sub(mi, full)));
            st[i] = mi;
        }
        q_external(st);
        for (int i = 0; i < 16; i++) st[i] = q_pow5(q_add(st[i], e.smp(rc1 + i)));
        q_external(st);
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_full, q_sub(e.smp(out + i), st[i])));
        for (int i = 0; i < 16; i++) st[i] = e.smp(in + i);
#pragma unroll 1
        for (int r = 0; r < 14; r++) {
            QM31 v = q_pow5(q_add(st[0], e.smp(rc0 + r)));
            QM31 mi = e.smp(mid + r);
            e.constraint(q_mul(is_partial, q_sub(mi, v)));
            st[0] = mi;
            q_internal(st);
        }
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_partial, q_sub(e.smp(out + i), st[i])));
        QM31 ext1 = e.smp(pre + 36), ext2 = e.smp(pre + 37), ext1_nz = e.smp(pre + 38), ext2_nz = e.smp(pre + 39);
        QM31 in_left = q_dbl(round_id), in_right = q_add(in_left, one), out_left = q_add(in_right, one),
             out_right = q_add(out_left, one);
#define EF4(base) q_combine_ef(e.smp(base), e.smp((base) + 1), e.smp((base) + 2), e.smp((base) + 3))
        e.relation(q_sub(q_mul(ext1_nz, is_first), not_first), q_add(q_mul(is_first, ext1), q_mul(not_first, in_left)),
                   EF4(in), EF4(in + 4));
        e.relation(q_sub(q_mul(ext2_nz, is_first), not_first), q_add(q_mul(is_first, ext2), q_mul(not_first, in_right)),
                   EF4(in + 8), EF4(in + 12));
        e.relation(q_add(q_mul(ext1_nz, is_last), not_last), q_add(q_mul(is_last, ext1), q_mul(not_last, out_left)),
                   EF4(out), EF4(out + 4));
        e.relation(q_add(q_mul(ext2_nz, is_last), not_last), q_add(q_mul(is_last, ext2), q_mul(not_last, out_right)),
                   EF4(out + 8), EF4(out + 12));
#undef EF4
        e.relation(q_mul(is_first, not_last), swap_val, e.smp(rc0));
        e.finalize_logup(3);
    }
    {  // composition/src/lib.rs:106-120
        QM31 left = q_combine_ef(e.smp(S_T3), e.smp(S_T3 + 1), e.smp(S_T3 + 2), e.smp(S_T3 + 3));
        QM31 right = q_combine_ef(e.smp(S_T3 + 4), e.smp(S_T3 + 5), e.smp(S_T3 + 6), e.smp(S_T3 + 7));
        uint32_t bound = umax(m.lp + 2, m.lq + 3);
        QM31 expected = q_add(left, q_mul(right, q_double_x(ox, bound - 2)));
        if (!q_eq(e.acc, expected)) flags |= 1u << R_COMPOSITION;
    }
    if (flags) atomicOr(&c.flags, flags);
}

// ------------------------------------------------------------------- k_plan
// Sorts the query positions, derives the decommitment plan (who owns which
// sibling, which witness index each lane consumes; see layout.hpp) and the
// per-proof constants of the DEEP quotients
// (components/recursive/answer/src/data_structures.rs:132-189).
// The per-query stages address their workspace by SLOT (position inside the current launch) and the
// per-proof records / the blob by PROOF index: proof = ids ? ids[slot] : p0 + slot.  A batch of mixed shapes
// is bucketed by n_queries on the host so that every launch uses G = that bucket's n_queries lanes per proof.
struct PlanPtrs {
    PlanHdr* hdr;
    uint32_t* ent;  // [slots][(maxM+1) * G]
    uint32_t* fl;   // [slots][2 * G]
    uint32_t G, maxM;
    const uint32_t* ids;  // slot -> proof index (nullptr: proof = p0 + slot)
    uint32_t p0;
    __device__ uint32_t proof_of(uint32_t slot) const { return ids ? ids[slot] : p0 + slot; }
};

__device__ inline int sample_index(int t, int col, int s) {
    if (t == 0) return S_T0 + col;
    if (t == 1) return S_T1 + col;
    if (t == 3) return S_T3 + col;
    return S_T2 + (col < 4 ? col : col < 8 ? 4 + 2 * (col - 4) + s : col < 12 ? 12 + (col - 8) : 16 + 2 * (col - 12) + s);
}

__global__ __launch_bounds__(64) void k_plan(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                             uint32_t n, const ProofMeta* __restrict__ metas,
                                             ProofCtx* __restrict__ ctxs, PlanPtrs pl) {
    __shared__ uint32_t sq[MAXQ][64];
    __shared__ uint8_t sp[MAXQ][64];
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const uint32_t p = pl.proof_of(slot);
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t nq = m.nq, M = m.M, A = m.A, B = m.B, G = pl.G;
    uint32_t flags = 0;
    // query positions (primitives/query/src/lib.rs:19-38), sorted ascending.  The sorted list is walked
    // M times below, so it lives in LDS (k-major: the 64 lanes of the block hit 64 different banks).
#define SQ(k) sq[(k)][threadIdx.x]
    for (uint32_t j = 0; j < nq; j++) {
        uint32_t v = c.raw_q[j] & ((1u << M) - 1u);
        uint32_t k = j;
        while (k > 0 && SQ(k - 1) > v) { SQ(k) = SQ(k - 1); sp[k][threadIdx.x] = sp[k - 1][threadIdx.x]; k--; }
        SQ(k) = v;
        sp[k][threadIdx.x] = (uint8_t)j;
    }
    for (uint32_t j = 0; j < nq; j++) { c.q[j] = SQ(j); c.qperm[j] = sp[j][threadIdx.x]; }
    for (uint32_t j = 0; j + 1 < nq; j++)
        if (SQ(j) == SQ(j + 1)) flags |= 1u << R_DUP_QUERY;  // answer/src/lib.rs:190-195
    // column log sizes, descending
    uint32_t n_sizes = 0;
    c.sizes[n_sizes++] = M;
    if (A == B) c.sizes[n_sizes++] = A;
    else { c.sizes[n_sizes++] = umax(A, B); c.sizes[n_sizes++] = umin(A, B); }
    c.n_sizes = n_sizes;
    if (n_sizes < 3) c.sizes[2] = 0;

    PlanHdr& h = pl.hdr[slot];
    uint32_t* ent = pl.ent + (size_t)slot * (pl.maxM + 1) * G;
    uint32_t* fl = pl.fl + (size_t)slot * 2 * G;
    // generic tables, node level l = M .. 1 (children of level l-1)
    uint32_t suffix = 0;
    h.lvl[M + 1] = 0;
    for (uint32_t l = M; l >= 1; l--) {
        uint32_t sh = M - l;  // node = q >> sh
        uint32_t k = 0, nodes_before = 0, runs_lacking = 0;
        while (k < nq) {
            uint32_t a = k;
            int split = -1;
            while (k + 1 < nq) {
                uint32_t x = SQ(k) ^ SQ(k + 1);
                int d = x ? 31 - __clz(x) : -1;
                if (d > (int)sh) break;
                if (d == (int)sh) split = (int)k;
                k++;
            }
            uint32_t bnd = k;
            k++;
            bool both = split >= 0;
            for (uint32_t j = a; j <= bnd; j++) {
                bool right = both && (int)j > split;
                uint32_t rb = nodes_before + (right ? 1u : 0u);
                uint32_t sib = both ? (right ? (uint32_t)split : (uint32_t)split + 1u) : 0xFFu;
                ent[l * G + j] = rb | (runs_lacking << 8) | (sib << 16);
            }
            nodes_before += both ? 2u : 1u;
            runs_lacking += both ? 0u : 1u;
        }
        suffix += runs_lacking;
        h.lvl[l] = nodes_before | (runs_lacking << 8) | (suffix << 16);
    }
    for (uint32_t j = 0; j < nq; j++) ent[j] = 0xFFu << 16;
    h.lvl[0] = 1u | (suffix << 16);
    // first-layer fri_witness bases (components/hints/src/folding.rs:414-451)
    {
        uint32_t base = 0;
        for (uint32_t g = 0; g < n_sizes; g++) {
            c.fw_base[g] = base;
            base += (h.lvl[c.sizes[g]] >> 8) & 0xFFu;
        }
        if (base != m.first.wit_n) flags |= 1u << R_FRI_FIRST;
    }
    // first-layer pair tree hash-witness plan (components/hints/src/folding.rs:107-206)
    {
        uint32_t wcount = 0, dslot = 0;
        for (uint32_t l = M; l-- > 0;) {
            h.wf[l + 1] = (uint16_t)wcount;
            bool child_data = false, data = false;
            for (uint32_t g = 0; g < n_sizes; g++) { child_data |= c.sizes[g] == l + 1; data |= c.sizes[g] == l; }
            uint32_t sh = M - l;
            uint32_t k = 0;
            while (k < nq) {
                uint32_t a = k;
                uint32_t node = SQ(k) >> sh;
                bool has_both_children = false;
                while (k + 1 < nq && (SQ(k + 1) >> sh) == node) {
                    if (((SQ(k) >> (sh - 1)) ^ (SQ(k + 1) >> (sh - 1))) & 1u) has_both_children = true;
                    k++;
                }
                uint32_t bnd = k;
                k++;
                uint32_t lack = (!child_data && !has_both_children) ? 1u : 0u;
                if (data) {
                    bool sib_present = (a > 0 && (SQ(a - 1) >> sh) == (node ^ 1u)) ||
                                       (bnd + 1 < nq && (SQ(bnd + 1) >> sh) == (node ^ 1u));
                    uint32_t w_self = 0xFFFFu, w_sib = 0xFFFFu;
                    if (node & 1u) {
                        if (!sib_present) { w_sib = wcount; wcount += 2; }
                        if (lack) { w_self = wcount; wcount += 1; }
                    } else {
                        if (lack) { w_self = wcount; wcount += 1; }
                        if (!sib_present) { w_sib = wcount; wcount += 2; }
                    }
                    if (dslot < 2)
                        for (uint32_t j = a; j <= bnd; j++) fl[dslot * G + j] = w_self | (w_sib << 16);
                } else {
                    wcount += lack;
                }
            }
            if (data) dslot++;
        }
        h.wf[0] = (uint16_t)wcount;
        h.wf_total = (uint16_t)umin(wcount, 0xFFFFu);
    }
    if (flags) atomicOr(&c.flags, flags);
#undef SQ
}

// ------------------------------------------------------------------ k_plan_par
// The same tables as k_plan, computed with one lane per (proof, query) like the other per-query kernels instead of
// one lane per proof (whose serial walk over LDS costs ~0.2 ms of pure latency per launch).  Everything follows
// from ONE family of bitmasks per proof: F[l] has bit j set when sorted query j is the first lane of a distinct
// node at tree level l (node = q >> (M - l)).  With N_l(x) = popcount(F[l] & bits[0..x]):
//   distinct nodes left of lane j at level l              N_l(j) - 1
//   a parent (level l-1 node, lanes s..e) has both children   N_l(e) - N_l(s) == 1; the right child starts at the
//                                                          one bit of F[l] & ~F[l-1] inside (s, e]
//   parents left of s that lack a child                    2 * popc(F[l-1] & below(s)) - popc(F[l] & below(s))
// The first-layer pair tree adds per-level witness weights (see k_plan); their prefix sums over the nodes of a level
// are popcounts of the same masks, and the running total over levels is a 30-step scan done by one lane.
struct M128 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ M128 m128_below(uint32_t x) {  // bits [0, x), x <= 128
    M128 r;
    r.lo = x >= 64 ? ~0ull : ((1ull << x) - 1ull);
    r.hi = x <= 64 ? 0ull : (x >= 128 ? ~0ull : ((1ull << (x - 64)) - 1ull));
    return r;
}
__device__ __forceinline__ M128 m128_and(M128 a, M128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ M128 m128_andn(M128 a, M128 b) { return {a.lo & ~b.lo, a.hi & ~b.hi}; }
__device__ __forceinline__ uint32_t m128_pop(M128 a) { return (uint32_t)(__popcll(a.lo) + __popcll(a.hi)); }
__device__ __forceinline__ uint32_t m128_popbelow(M128 a, uint32_t x) { return m128_pop(m128_and(a, m128_below(x))); }
// highest set bit at or below x (the mask has bit 0 set), lowest set bit above x or `none`
__device__ __forceinline__ uint32_t m128_last_le(M128 a, uint32_t x) {
    M128 t = m128_and(a, m128_below(x + 1));
    return t.hi ? 127u - (uint32_t)__clzll((long long)t.hi) : 63u - (uint32_t)__clzll((long long)t.lo);
}
__device__ __forceinline__ uint32_t m128_first_gt(M128 a, uint32_t x, uint32_t none) {
    M128 t = m128_andn(a, m128_below(x + 1));
    if (t.lo) return (uint32_t)__ffsll((long long)t.lo) - 1u;
    if (t.hi) return 63u + (uint32_t)__ffsll((long long)t.hi);
    return none;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_plan_par(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                    uint32_t n, const ProofMeta* __restrict__ metas,
                                                    ProofCtx* __restrict__ ctxs, PlanPtrs pl) {
    __shared__ unsigned long long F[64][32][2];   // per_block <= 64 proofs, levels 0..30
    __shared__ uint32_t raw[BLOCK], sq[BLOCK];
    __shared__ uint8_t sp[BLOCK];
    __shared__ uint32_t tl[64][32];               // per level: nodes | lacking << 8
    __shared__ uint32_t tw[64][32];               // per node level: witness weight of the first-layer pair tree
    __shared__ uint32_t wsum[64][32];             // wf[l + 1]
    const uint32_t G = pl.G, per_block = BLOCK / G;
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * per_block + grp;
    bool livep = grp < per_block && slot < n;
    const uint32_t p = livep ? pl.proof_of(slot) : 0u;
    const ProofMeta* m = livep ? &metas[p] : nullptr;
    livep = livep && m->reason == R_OK;
    const uint32_t nq = livep ? m->nq : 0u, M = livep ? m->M : 1u, A = livep ? m->A : 0u, B = livep ? m->B : 0u;
    const bool live = livep && j < nq;
    ProofCtx* c = livep ? &ctxs[p] : nullptr;
    const uint32_t gbase = grp * G;
    for (uint32_t i = threadIdx.x; i < 64u * 32u * 2u; i += BLOCK) (&F[0][0][0])[i] = 0ull;
    const uint32_t v0 = live ? (c->raw_q[j] & ((1u << M) - 1u)) : 0xFFFFFFFFu;
    raw[threadIdx.x] = v0;
    __syncthreads();
    // rank sort (primitives/query/src/lib.rs:19-38): ties broken by transcript index, as the insertion sort does
    if (live) {
        uint32_t rank = 0;
        for (uint32_t k = 0; k < nq; k++) {
            const uint32_t vk = raw[gbase + k];
            rank += (vk < v0 || (vk == v0 && k < j)) ? 1u : 0u;
        }
        sq[gbase + rank] = v0;
        sp[gbase + rank] = (uint8_t)j;
    }
    __syncthreads();
    uint32_t flags = 0;
    const uint32_t v = live ? sq[gbase + j] : 0u;
    if (live) {
        c->q[j] = v;
        c->qperm[j] = sp[gbase + j];
        if (j + 1 < nq && sq[gbase + j + 1] == v) flags |= 1u << R_DUP_QUERY;  // answer/src/lib.rs:190-195
        // lane j starts a new node at every level l >= M - (highest bit in which it differs from lane j-1)
        uint32_t lstart = 0;
        if (j > 0) {
            const uint32_t x = v ^ sq[gbase + j - 1];
            lstart = x ? M - (31u - (uint32_t)__clz((int)x)) : M + 1u;
        }
        for (uint32_t l = lstart; l <= M; l++) atomicOr(&F[grp][l][j >> 6], 1ull << (j & 63u));
    }
    __syncthreads();
    auto mask = [&](uint32_t l) { return M128{F[grp][l][0], F[grp][l][1]}; };
    // column log sizes, descending
    uint32_t sizes[3] = {M, A == B ? A : umax(A, B), A == B ? 0u : umin(A, B)};
    const uint32_t n_sizes = A == B ? 2u : 3u;
    auto is_size = [&](uint32_t l) { return l == sizes[0] || l == sizes[1] || (n_sizes == 3 && l == sizes[2]); };
    // per-level totals, levels dealt round-robin to the proof's lanes
    if (live) {
        for (uint32_t l = j; l <= M; l += nq) {
            const M128 Fl = mask(l);
            const uint32_t nodes = m128_pop(Fl);
            uint32_t lacking = 0;
            if (l >= 1) lacking = 2u * m128_pop(mask(l - 1)) - nodes;
            tl[grp][l] = nodes | (lacking << 8);
            if (l < M) {
                const uint32_t both_total = m128_pop(m128_andn(mask(l + 1), Fl));
                const uint32_t nosib_total = l == 0 ? 1u : m128_pop(mask(l - 1)) - m128_pop(m128_andn(Fl, mask(l - 1)));
                tw[grp][l] = (is_size(l + 1) ? 0u : nodes - both_total) + (is_size(l) ? 2u * nosib_total : 0u);
            }
        }
    }
    __syncthreads();
    PlanHdr* h = livep ? &pl.hdr[slot] : nullptr;
    if (live && j == 0) {
        uint32_t suffix = 0;
        h->lvl[M + 1] = 0;
        for (uint32_t l = M; l >= 1; l--) {
            suffix += (tl[grp][l] >> 8) & 0xFFu;
            h->lvl[l] = tl[grp][l] | (suffix << 16);
        }
        h->lvl[0] = 1u | (suffix << 16);
        uint32_t W = 0;
        for (uint32_t l = M; l-- > 0;) {
            wsum[grp][l] = W;
            h->wf[l + 1] = (uint16_t)W;
            W += tw[grp][l];
        }
        h->wf[0] = (uint16_t)W;
        h->wf_total = (uint16_t)umin(W, 0xFFFFu);
        c->n_sizes = n_sizes;
        c->sizes[0] = sizes[0]; c->sizes[1] = sizes[1]; c->sizes[2] = n_sizes == 3 ? sizes[2] : 0u;
        // first-layer fri_witness bases (components/hints/src/folding.rs:414-451)
        uint32_t base = 0;
        for (uint32_t g = 0; g < n_sizes; g++) {
            c->fw_base[g] = base;
            base += (tl[grp][sizes[g]] >> 8) & 0xFFu;
        }
        if (base != m->first.wit_n) flags |= 1u << R_FRI_FIRST;
    }
    __syncthreads();
    if (live) {
        uint32_t* ent = pl.ent + (size_t)slot * (pl.maxM + 1) * G;
        uint32_t* fl = pl.fl + (size_t)slot * 2 * G;
        ent[j] = 0xFFu << 16;
        for (uint32_t l = 1; l <= M; l++) {
            const M128 Fl = mask(l), Fp = mask(l - 1);
            const uint32_t s = m128_last_le(Fp, j);                     // first lane of my parent's run
            const uint32_t e = m128_first_gt(Fp, j, nq) - 1u;           // its last lane
            const uint32_t Ns = m128_popbelow(Fl, s + 1), Ne = m128_popbelow(Fl, e + 1), Nj = m128_popbelow(Fl, j + 1);
            const bool both = Ne - Ns == 1u;
            const bool right = both && Nj - Ns == 1u;
            const uint32_t second = both ? m128_first_gt(Fl, s, nq) : 0u;  // first lane of the right child
            const uint32_t sib = both ? (right ? second - 1u : second) : 0xFFu;
            const uint32_t lack_before = 2u * m128_popbelow(Fp, s) - m128_popbelow(Fl, s);
            ent[l * G + j] = (Nj - 1u) | (lack_before << 8) | (sib << 16);
        }
        // first-layer pair tree: witness indices at the (up to two) non-leaf column levels (folding.rs:107-206)
        for (uint32_t d = 0; d + 1 < n_sizes; d++) {
            const uint32_t l = sizes[1 + d];
            if (l >= M) continue;
            const M128 Fl = mask(l), Fc = mask(l + 1);
            const uint32_t f = m128_last_le(Fl, j), e = m128_first_gt(Fl, j, nq) - 1u;
            const bool has_both = m128_popbelow(Fc, e + 1) - m128_popbelow(Fc, f + 1) == 1u;
            const bool child_data = is_size(l + 1);
            const bool lack = !child_data && !has_both;
            bool sib_present = false;
            uint32_t nosib_before = 0;
            if (l >= 1) {
                const M128 Fp = mask(l - 1);
                const uint32_t s = m128_last_le(Fp, j), pe = m128_first_gt(Fp, j, nq) - 1u;
                sib_present = m128_popbelow(Fl, pe + 1) - m128_popbelow(Fl, s + 1) == 1u;
                nosib_before = m128_popbelow(Fp, s) - m128_popbelow(m128_andn(Fl, Fp), s);
            }
            const uint32_t nodes_before = m128_popbelow(Fl, f), both_before = m128_popbelow(m128_andn(Fc, Fl), f);
            const uint32_t base = wsum[grp][l] + (child_data ? 0u : nodes_before - both_before) + 2u * nosib_before;
            const bool odd = (v >> (M - l)) & 1u;
            uint32_t w_self = 0xFFFFu, w_sib = 0xFFFFu;
            if (odd) {
                if (!sib_present) w_sib = base;
                if (lack) w_self = base + (sib_present ? 0u : 2u);
            } else {
                if (lack) w_self = base;
                if (!sib_present) w_sib = base + (lack ? 1u : 0u);
            }
            fl[d * G + j] = (w_self & 0xFFFFu) | (w_sib << 16);
        }
    }
    if (flags) atomicOr(&c->flags, flags);
}

// ------------------------------------------------------------------ k_export_transcript
// One lane per output word: ProofCtx -> the flat row layout of include/rsv.h (RSV_TRANSCRIPT_WORDS).
constexpr uint32_t TR_WORDS = 40 + 4 * (MAX_INNER + 1) + MAXQ;
__global__ __launch_bounds__(256) void k_export_transcript(uint32_t n, const ProofMeta* __restrict__ metas,
                                                            const ProofCtx* __restrict__ ctxs, uint32_t* __restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n * TR_WORDS) return;
    const uint32_t p = (uint32_t)(gid / TR_WORDS), k = (uint32_t)(gid % TR_WORDS);
    const ProofMeta& m = metas[p];
    const ProofCtx& c = ctxs[p];
    uint32_t v = 0;
    if (m.reason != R_OK) v = k == 0 ? (uint32_t)R_PARSE : 0u;
    else if (k == 0) v = (c.flags & (1u << R_POW)) ? (uint32_t)R_POW : (uint32_t)R_OK;
    else if (k == 1) v = m.n_inner + 1;
    else if (k == 2) v = m.nq;
    else if (k == 3) v = m.M;
    else if (k < 8) v = c.z[k - 4];
    else if (k < 12) v = c.alpha[k - 8];
    else if (k < 16) v = c.rc[k - 12];
    else if (k < 20) v = c.oods_t[k - 16];
    else if (k < 24) v = c.oods_x[k - 20];
    else if (k < 28) v = c.oods_y[k - 24];
    else if (k < 32) v = c.after[k - 28];
    else if (k < 40) v = c.pow_digest[k - 32];
    else if (k < 40 + 4 * (MAX_INNER + 1)) { uint32_t a = (k - 40) >> 2; v = a <= m.n_inner ? c.fri_alpha[a][(k - 40) & 3] : 0u; }
    else { uint32_t q = k - (40 + 4 * (MAX_INNER + 1)); v = q < m.nq ? c.raw_q[q] : 0u; }
    out[gid] = v;
}

// ------------------------------------------------------------------ k_qconst
// One lane per proof: the query-independent constants of the DEEP quotients — alpha powers and, per column
// log size and sample point, the summed line coefficients.  Needs only the transcript, so it runs on the side
// stream next to k_plan (whose tables need only the query positions).
__global__ __launch_bounds__(64) void k_qconst(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                               uint32_t n, const ProofMeta* __restrict__ metas,
                                               ProofCtx* __restrict__ ctxs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    const uint32_t M = m.M, A = m.A, B = m.B;
    // column log sizes, descending (the same list k_plan stores in ProofCtx::sizes)
    uint32_t sizes[3] = {M, A == B ? A : umax(A, B), A == B ? 0u : umin(A, B)};
    const uint32_t n_sizes = A == B ? 2u : 3u;
    // quotient constants: alpha_k = -2u * after^k (data_structures.rs:162-189)
    QM31 after = ldq(c.after);
    {
        QM31 ak = q_mk(0, 0, m_neg(2), 0);
#pragma unroll 1
        for (int k = 0; k < N_APOW; k++) { stq(c.apow[k], ak); ak = q_mul(ak, after); }
    }
    QM31 ox = ldq(c.oods_x), oy = ldq(c.oods_y);
    for (uint32_t g = 0; g < n_sizes; g++) {
        uint32_t l = sizes[g];
        // batch 0: OODS point; batch 1: OODS - g_{component log size} (answer/src/lib.rs:62-72)
        uint32_t comp_log = (l == A) ? m.lp : m.lq;
        CPoint step = cp_gen_mul(1u << (31u - comp_log));
        step.y = m_neg(step.y);
        QM31 sx = q_sub(q_mul_m(ox, step.x), q_mul_m(oy, step.y));
        QM31 sy = q_add(q_mul_m(ox, step.y), q_mul_m(oy, step.x));
        uint32_t k_run = 0, n_batches = (l == M) ? 1u : 2u;
        for (uint32_t bi = 0; bi < n_batches; bi++) {
            QM31 px = bi ? sx : ox, py = bi ? sy : oy;
            QM31 sa = q_zero(), sb = q_zero();
            for (int t = 0; t < 4; t++) {
                uint32_t c0, c1;
                if (l == M) { if (t != 3) continue; c0 = 0; c1 = 8; }
                else {
                    if (t == 3) continue;
                    c0 = (l == A) ? 0u : plonk_cols(t);
                    c1 = (l == B) ? tree_cols(t) : plonk_cols(t);
                }
                for (uint32_t col = c0; col < c1; col++) {
                    uint32_t ns = n_samples_of(t, (int)col);
                    if (bi == 1 && ns != 2) continue;
                    int si = sample_index(t, (int)col, bi == 1 ? 0 : (int)ns - 1);
                    QM31 v = ldq(w + SAMPLES.off[si]);
                    QM31 ak = ldq(c.apow[k_run++]);
                    // complex_conjugate_line_coeffs_var (data_structures.rs:132-160)
                    sa = q_add(sa, q_mul_c(ak, v.b));
                    sb = q_add(sb, q_mul_c(ak, c_sub(c_mul(v.a, py.b), c_mul(v.b, py.a))));
                }
            }
            QBatch& qb = c.batch[g][bi];
            stq(qb.sa, sa); stq(qb.sb, sb);
            qb.prx[0] = px.a.a; qb.prx[1] = px.a.b; qb.pix[0] = px.b.a; qb.pix[1] = px.b.b;
            qb.pry[0] = py.a.a; qb.pry[1] = py.a.b; qb.piy[0] = py.b.a; qb.piy[1] = py.b.b;
        }
        c.n_batches[g] = n_batches;
    }
}

// ------------------------------------------------------------------ k_query
// One lane per (proof, query): DEEP quotients for every column log size
// (answer/src/lib.rs:260-315,356-382), the circle->line fold of the first FRI
// layer (folding/src/lib.rs:57-90), the line folds of the inner layers
// (:120-192) and the last-layer polynomial check (:194-204).  Values owned by
// other queries of the same proof (pair siblings) are exchanged through LDS.
// Writes, for the Merkle kernels, the (self, sibling) leaf values of every FRI tree.
struct QueryArgs {
    const uint8_t* blob;
    const uint64_t* offsets;
    uint32_t n;
    const ProofMeta* metas;
    ProofCtx* ctxs;
    PlanPtrs pl;
    uint32_t* leafv;  // [n][3 + maxInner][G][8]
    uint32_t maxInner;
    uint32_t* folded_out;  // optional [n][3][G][4]: first-layer folds per size group, transcript query order
};

__device__ __forceinline__ uint32_t ent_rb(uint32_t e) { return e & 0xFFu; }
__device__ __forceinline__ uint32_t ent_lb(uint32_t e) { return (e >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t ent_sib(uint32_t e) { return (e >> 16) & 0xFFu; }
__device__ __forceinline__ uint32_t lvl_nd(uint32_t v) { return v & 0xFFu; }
__device__ __forceinline__ uint32_t lvl_tl(uint32_t v) { return (v >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t lvl_s(uint32_t v) { return v >> 16; }

__device__ inline QM31 fold_pair(QM31 self, QM31 sib, bool odd, uint32_t inv_coord, QM31 alpha) {
    QM31 l = odd ? sib : self, r = odd ? self : sib;
    return q_add(q_add(l, r), q_mul(q_mul_m(q_sub(l, r), inv_coord), alpha));
}

// LinePolyVar::eval_at_point (primitives/line/src/lib.rs:39-67): fold(coeffs, [x, pi(x), pi(pi(x)), ...]) with
// fold(v, [f, rest]) = fold(v_lo, rest) + f * fold(v_hi, rest), i.e. sum_i coeff_i * prod_k d[k]^(bit (log_n-1-k) of i).
// The weights factor into a table over the low 4 index bits (registers) times a product over the high bits.
// cf: n = 2^log_n QM31 coefficients (4 words each).
__device__ inline QM31 line_eval(const uint32_t* __restrict__ cf, uint32_t log_n, uint32_t n, uint32_t x) {
    uint32_t d[16];
    for (uint32_t k = 0; k < 16; k++) { d[k] = (k < log_n) ? x : 1u; x = m_sub(m_dbl(m_sqr(x)), 1u); }
    const uint32_t nlo = log_n < 4 ? log_n : 4u;
    uint32_t wl[16];
#pragma unroll
    for (int t = 0; t < 16; t++) wl[t] = 1u;
    // low index bit b pairs with d[log_n - 1 - b]
#pragma unroll
    for (int b = 0; b < 4; b++) {
        uint32_t db = (uint32_t)b < nlo ? d[(log_n - 1 - b) & 15u] : 1u;
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (t & (1 << b)) wl[t] = m_mul(wl[t], db);
    }
    QM31 acc = q_zero();
    const uint32_t n_hi = n >> nlo, n_lo = 1u << nlo;
#pragma unroll 1
    for (uint32_t hi = 0; hi < n_hi; hi++) {
        uint32_t wh = 1u;
        for (uint32_t b = 0; b + nlo < log_n; b++)
            if ((hi >> b) & 1u) wh = m_mul(wh, d[(log_n - 1 - nlo - b) & 15u]);
        QM31 inner = q_zero();
#pragma unroll
        for (int t = 0; t < 16; t++)
            if ((uint32_t)t < n_lo) inner = q_add(inner, q_mul_m(ldq(cf + 4 * ((hi << nlo) + t)), wl[t]));
        acc = q_add(acc, q_mul_m(inner, wh));
    }
    return acc;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_query(QueryArgs a) {
    __shared__ uint32_t xq[BLOCK][4];
    const uint32_t G = a.pl.G, per_block = BLOCK / G;
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * per_block + grp;
    bool live = grp < per_block && slot < a.n;
    const uint32_t p = live ? a.pl.proof_of(slot) : 0u;
    const ProofMeta* m = live ? &a.metas[p] : nullptr;
    live = live && m->reason == R_OK && j < m->nq;
    ProofCtx* c = live ? &a.ctxs[p] : nullptr;
    const uint32_t* w = live ? reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]) : nullptr;
    const uint32_t* ent = live ? a.pl.ent + (size_t)slot * (a.pl.maxM + 1) * G : nullptr;
    const PlanHdr* h = live ? &a.pl.hdr[slot] : nullptr;
    uint32_t* leafv = live ? a.leafv + ((size_t)slot * (3 + a.maxInner)) * G * 8 : nullptr;
    const uint32_t gbase = grp * G;
    uint32_t flags = 0;
    uint32_t M = live ? m->M : 0, A = live ? m->A : 0, B = live ? m->B : 0;
    uint32_t qj = live ? c->q[j] : 0;
    uint32_t n_sizes = live ? c->n_sizes : 0;
    QM31 first[3];
    // Domain points.  One scalar multiplication gives the point of the query at level M; the points at the
    // smaller column sizes follow by the doubling map pi(x, y) = (2x^2 - 1, 2xy): doubling the level-l point of
    // position pos gives the level-(l-1) point of pos >> 1 up to the sign of y, which is fixed by bit 0 of the
    // respective positions (CanonicCoset::circle_domain().at(bit_reverse(.))).
    CPoint dp[3];
    if (live) {
        CPoint cur = domain_point(M, qj);
        uint32_t lvl = M;
        for (uint32_t g = 0; g < n_sizes; g++) {
            const uint32_t l = c->sizes[g];
            while (lvl > l) {
                const uint32_t pos = qj >> (M - lvl);
                uint32_t y2 = m_dbl(m_mul(cur.x, cur.y));
                cur.x = m_sub(m_dbl(m_sqr(cur.x)), 1u);
                cur.y = ((pos ^ (pos >> 1)) & 1u) ? m_neg(y2) : y2;
                lvl--;
            }
            dp[g] = cur;
        }
    }
    // ---- DEEP quotients + first-layer fold, per column log size
    for (uint32_t g = 0; g < 3; g++) {
        QM31 answer = q_zero();
        uint32_t l = 0, pos = 0;
        bool on = live && g < n_sizes;
        if (on) {
            l = c->sizes[g];
            pos = qj >> (M - l);
            QM31 r0 = q_zero(), r1 = q_zero();
            uint32_t col = 0, dbl = 0;
            const uint32_t ncols_group = (l == M) ? 8u : ((l == A ? 30u : 0u) + (l == B ? 96u : 0u));
            for (int t = 0; t < 4; t++) {
                if ((l == M) != (t == 3)) continue;
                const uint32_t mx = (t == 3) ? M : umax(A, B);
                const uint32_t nc_leaf = (t == 3) ? 8u : ((A == mx ? plonk_cols(t) : 0u) + (B == mx ? poseidon_cols(t) : 0u));
                const uint32_t* qv = w + m->qv_off[t];
                const uint32_t qv_n = m->qv_n[t];
                // the two components' columns of this tree at level l
                for (int comp = 0; comp < 2; comp++) {
                    uint32_t cl = (t == 3) ? M : (comp == 0 ? A : B);
                    if (cl != l) continue;
                    uint32_t nc = (t == 3) ? (comp == 0 ? 8u : 0u) : (comp == 0 ? plonk_cols(t) : poseidon_cols(t));
                    if (nc == 0) continue;
                    uint32_t off;
                    if (cl == mx) off = ent_rb(ent[mx * G + j]) * nc_leaf + ((comp == 1 && A == B && t != 3) ? plonk_cols(t) : 0u);
                    else off = lvl_nd(h->lvl[mx]) * nc_leaf + ent_rb(ent[cl * G + j]) * nc;
                    bool inb = off + nc <= qv_n;
                    if (!inb) flags |= 1u << (R_MERKLE_T0 + t);
                    for (uint32_t k = 0; k < nc; k++) {
                        uint32_t v = inb ? qv[off + k] : 0u;
                        r0 = q_add(r0, q_mul_m(ldq(c->apow[col]), v));
                        if (t == 2 && (k & 4)) {
                            r1 = q_add(r1, q_mul_m(ldq(c->apow[ncols_group + dbl]), v));
                            dbl++;
                        }
                        col++;
                    }
                }
            }
            for (uint32_t bi = 0; bi < c->n_batches[g]; bi++) {
                const QBatch& qb = c->batch[g][bi];
                CM31 prx = c_mk(qb.prx[0], qb.prx[1]), pix = c_mk(qb.pix[0], qb.pix[1]);
                CM31 pry = c_mk(qb.pry[0], qb.pry[1]), piy = c_mk(qb.piy[0], qb.piy[1]);
                QM31 num = q_sub(q_mul_c(bi ? r1 : r0, piy), q_add(q_mul_m(ldq(qb.sa), dp[g].y), ldq(qb.sb)));
                CM31 den = c_sub(c_mul(c_sub(prx, c_mk(dp[g].x, 0)), piy), c_mul(c_sub(pry, c_mk(dp[g].y, 0)), pix));
                answer = q_add(answer, q_mul_c(num, c_inv(den)));
            }
        }
        // exchange answers: the pair sibling may be another query of this proof
        stq(xq[threadIdx.x], answer);
        __syncthreads();
        if (on) {
            uint32_t e = ent[l * G + j];
            QM31 sib;
            if (ent_sib(e) != 0xFFu) sib = ldq(xq[gbase + ent_sib(e)]);
            else {
                uint32_t wi = c->fw_base[g] + ent_lb(e);
                sib = wi < m->first.wit_n ? ldq(w + m->first.wit_off + 4 * wi) : q_zero();
            }
            uint32_t* lv = leafv + ((size_t)g * G + j) * 8;
            stq(lv, answer); stq(lv + 4, sib);
            // fold circle -> line with 1/y of the pair's base point (folding/src/lib.rs:57-90); the base
            // point (bit 0 of the position cleared) is the conjugate of this point when the position is odd
            uint32_t by = (pos & 1u) ? m_neg(dp[g].y) : dp[g].y;
            first[g] = fold_pair(answer, sib, pos & 1u, m_inv(by), ldq(c->fri_alpha[M - l]));
            if (a.folded_out) stq(a.folded_out + (((size_t)slot * 3 + g) * G + c->qperm[j]) * 4, first[g]);
        }
        __syncthreads();
    }
    // ---- inner layers (folding/src/lib.rs:120-192)
    // x-coordinate of the pair base point at line-domain level l: X_{M-1} = +-x_M and X_{l-1} = +-(2 X_l^2 - 1)
    // (half_odds(l).at(i) doubles to half_odds(l-1).at(i mod 2^(l-1)); clearing bit 0 of an odd position moves
    // the bit-reversed index by half the coset = the point (-1, 0), i.e. negates x).
    QM31 folded = q_zero();
    uint32_t l = M;
    uint32_t X = live ? dp[0].x : 0u;
    for (uint32_t i = 0; i < a.maxInner; i++) {
        bool on = live && i < m->n_inner;
        if (on) {
            for (uint32_t g = 0; g < n_sizes; g++)
                if (c->sizes[g] == l) {
                    QM31 al = ldq(c->fri_alpha[i]);
                    folded = q_add(q_mul(q_mul(al, al), folded), first[g]);
                }
            l -= 1;
        }
        stq(xq[threadIdx.x], folded);
        __syncthreads();
        if (on) {
            uint32_t pos = qj >> (M - l);
            uint32_t e = ent[l * G + j];
            const FriLayerRef& L = m->inner[i];
            QM31 sib;
            if (ent_sib(e) != 0xFFu) sib = ldq(xq[gbase + ent_sib(e)]);
            else {
                uint32_t wi = ent_lb(e);
                sib = wi < L.wit_n ? ldq(w + L.wit_off + 4 * wi) : q_zero();
            }
            if (j == 0 && lvl_tl(h->lvl[l]) != L.wit_n) flags |= 1u << R_FRI_INNER;  // hints/src/folding.rs:558
            uint32_t* lv = leafv + ((size_t)(3 + i) * G + j) * 8;
            stq(lv, folded); stq(lv + 4, sib);
            uint32_t xr = (i == 0) ? X : m_sub(m_dbl(m_sqr(X)), 1u);
            X = (pos & 1u) ? m_neg(xr) : xr;
            folded = fold_pair(folded, sib, pos & 1u, m_inv(X), ldq(c->fri_alpha[i + 1]));
        }
        __syncthreads();
    }
    // ---- last layer (folding/src/lib.rs:194-204, primitives/line/src/lib.rs:39-67)
    if (live) {
        // x of half_odds(l-1).at(bit_reverse(pos >> 1)) = pi(X_l) (for a proof without inner layers: pi of x_M)
        QM31 acc = line_eval(w + m->last_off, m->log_last, m->last_n, m_sub(m_dbl(m_sqr(X)), 1u));
        if (!q_eq(acc, folded)) flags |= 1u << R_FRI_LAST;
    }
    if (flags) atomicOr(&c->flags, flags);
}

// ------------------------------------------------------------------ probes
// Batch probes of the arithmetic the verify kernels are made of (include/rsv.h: rsv_field_op, rsv_domain_points,
// rsv_line_eval): the SAME device functions, one lane per item, so that each can be checked on its own
// (SURVEY rows a1, a2, a8 and the last layer of a12).
__global__ __launch_bounds__(256) void k_field_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                   uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    QM31 x = ldq(a + 4 * i), y = b ? ldq(b + 4 * i) : q_zero(), r = q_zero();
    switch (op) {
        case 0: r = q_add(x, y); break;
        case 1: r = q_sub(x, y); break;
        case 2: r = q_mul(x, y); break;
        case 3: r = q_inv(x); break;
        case 4: r = q_mk(m_mul(x.a.a, y.a.a), 0, 0, 0); break;           // M31 product of the first words
        case 5: r = q_mk(m_inv(x.a.a), 0, 0, 0); break;                  // M31 inverse of the first word
        case 6: { CM31 c = c_mul(x.a, y.a); r = q_mk(c.a, c.b, 0, 0); } break;
        case 7: { CM31 c = c_inv(x.a); r = q_mk(c.a, c.b, 0, 0); } break;
        case 8: r = q_mul_i(x); break;
        case 9: r = q_mul_u(x); break;
        case 10: {                                                       // x^e, e = first word of y (QM31Var::pow)
            QM31 acc = q_one(), base = x;
            for (uint32_t e = y.a.a; e; e >>= 1) { if (e & 1u) acc = q_mul(acc, base); base = q_mul(base, base); }
            r = acc;
        } break;
        default: break;
    }
    stq(out + 4 * i, r);
}

__global__ __launch_bounds__(256) void k_domain_points(uint32_t log_size, const uint32_t* __restrict__ q,
                                                        uint32_t* __restrict__ xy, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    CPoint p = domain_point(log_size, q[i] & ((1u << log_size) - 1u));
    xy[2 * i] = p.x; xy[2 * i + 1] = p.y;
}

__global__ __launch_bounds__(256) void k_line_eval(const uint32_t* __restrict__ coeffs, uint32_t log_n,
                                                    const uint32_t* __restrict__ x, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stq(out + 4 * i, line_eval(coeffs, log_n, 1u << log_n, x[i]));
}

// --------------------------------------------------------------- k_row_hash
// Column hashing of the trace trees does not depend on the transcript: queried_values lists one row of
// column values per distinct queried node, in ascending node order, leaf level first.  The sponge over row r
// (primitives/merkle/src/lib.rs:50-181) can therefore run UNDERNEATH the latency-bound transcript kernel
// (side stream): one lane per (proof, tree, row) hashes leaf-level row r counted from the start of
// queried_values and lower-level row r counted from its END (where that block begins depends on how many
// leaves are distinct, which is only known once the queries are).  k_trace_merkle then just picks its rows.
//   rowh[((p*4 + t)*2 + 0)*G + r] = leaf hash of leaf row r
//   rowh[((p*4 + t)*2 + 1)*G + r] = column capacity digest of the r-th LAST lower-level row
struct RowHashArgs {
    const uint8_t* blob;
    const uint64_t* offsets;
    uint32_t n, G;         // slots in this launch, lanes per proof (= n_queries of the bucket)
    const ProofMeta* metas;
    uint32_t* rowh;        // [proof][4][2][Grow][8]
    uint32_t Grow;         // row stride of rowh (max n_queries of the batch)
    const uint32_t* ids;   // slot -> proof (nullptr: identity)
};

__global__ __launch_bounds__(256) void k_row_hash(RowHashArgs a) {
    RSV_TAG(2);
    const uint32_t G = a.G, per_block = 256 / G;
    const uint32_t grp = threadIdx.x / G, r = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * per_block + grp;
    const int t = blockIdx.y;
    if (grp >= per_block || slot >= a.n) return;
    const uint32_t p = a.ids ? a.ids[slot] : slot;
    const ProofMeta& m = a.metas[p];
    if (m.reason != R_OK || r >= m.nq) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
    const uint32_t A = m.A, B = m.B, M = m.M;
    const uint32_t mx = (t == 3) ? M : umax(A, B);
    const uint32_t nc_leaf = (t == 3) ? 8u : ((A == mx ? plonk_cols(t) : 0u) + (B == mx ? poseidon_cols(t) : 0u));
    const uint32_t nc_lower = (t == 3 || A == B) ? 0u : (A < B ? plonk_cols(t) : poseidon_cols(t));
    const uint32_t* qv = w + m.qv_off[t];
    const uint32_t qv_n = m.qv_n[t];
    uint32_t* out = a.rowh + (((size_t)p * 4 + t) * 2) * a.Grow * 8;
    if ((r + 1) * nc_leaf <= qv_n) store_hash(out + (size_t)r * 8, leaf_from_capacity(sponge_capacity(qv + r * nc_leaf, nc_leaf)));
    if (nc_lower && (r + 1) * nc_lower <= qv_n)
        store_hash(out + ((size_t)a.Grow + r) * 8, sponge_capacity(qv + qv_n - (r + 1) * nc_lower, nc_lower));
}

// ----------------------------------------------------------- k_trace_merkle
// SinglePathMerkleProofVar::verify (components/recursive/data_structures/src/lib.rs:315-354)
// for the four commitment trees: blockIdx.y = tree, one lane per (proof, query).
struct MerkleArgs {
    const uint8_t* blob;
    const uint64_t* offsets;
    uint32_t n;
    const ProofMeta* metas;
    ProofCtx* ctxs;
    PlanPtrs pl;
    const uint32_t* leafv;
    uint32_t maxInner;
    uint32_t Lc;  // cap level: levels below Lc are hashed by merkle_cap (0 = walk every path to the root)
    const uint32_t* rowh;  // k_row_hash output, [proof][4][2][Grow][8]
    uint32_t Grow;
    // optional per-query authentication paths of the trace trees, transcript query order:
    //   path_sib[((slot*4 + t)*G + i)*maxM + k]  = sibling hash at the k-th level above the leaf (8 words)
    //   path_pos[(slot*4 + t)*G + i]              = position of query i at the tree's leaf level
    uint32_t* path_sib;
    uint32_t* path_pos;
    //   path_cols[((slot*4 + t)*G + i)*64 + k]    = SinglePathMerkleProof::columns: the leaf-level column values of
    //                                               query i, then those at the lower column log size (may be null)
    uint32_t* path_cols;
    // optional per-query pair paths of the FRI trees (SinglePairMerkleProof, components/hints/src/folding.rs:214-287):
    //   pair_sib [(((slot*(1+maxInner) + s)*G + i)*maxM + k]  sibling_hashes[k] of tree s (0 = first layer), 8 words
    //   pair_cols[(((slot*(1+maxInner) + s)*G + i)*3 + c]     c-th data level from the top: self | sibling value, 8 words
    uint32_t* pair_sib;
    uint32_t* pair_cols;
};

// ---------------------------------------------------------------- merkle_cap
// Top of a tree.  Below level Lc (2^Lc <= queries per proof) the query paths of a proof have merged into
// at most 2^l distinct nodes per level, so continuing one-lane-per-path would hash every shared node up to
// n_queries times.  Here the lanes of the workgroup are re-dealt densely over (proof, node position): level
// l costs per_block * 2^l lanes instead of per_block * G.  Nodes live in LDS (xch, two buffers), presence
// in a per-proof bitmask; a missing child is the next hash_witness entry in ascending node order, exactly
// the batched order of components/hints/src/decommit.rs:91-139 and folding.rs:116-206.
struct CapGroup {
    const uint32_t* hw;    // hash witness of this (proof, tree)
    const uint32_t* lvl;   // PlanHdr::lvl
    const uint16_t* wf;    // PlanHdr::wf (first-layer pair tree) or nullptr
    const uint32_t* root;  // expected root (8 words)
    uint32_t* flags;       // ProofCtx::flags
    uint32_t hw_n, s_top, active, fail_bit;
};

// emit (optional, path emission): this lane's path buffer; the sibling consumed at child level l + 1 goes to entry
// emit_top - l (8 words each), so a query's path is complete although its lane does not walk the top levels.
template <int BLOCK>
__device__ __forceinline__ void merkle_cap(uint32_t (*xch)[BLOCK][8], unsigned long long (*mask)[64], CapGroup* grp_desc,
                                           uint32_t Lc, uint32_t per_block, bool live, uint32_t grp, uint32_t pos,
                                           const Hash8& cur, uint32_t* emit = nullptr, uint32_t emit_top = 0) {
    const uint32_t t = threadIdx.x;
    __syncthreads();  // xch is free, descriptors written
    if (t < per_block) { mask[0][t] = 0; mask[1][t] = 0; }
    __syncthreads();
    if (live) {
        store_hash(xch[0][(grp << Lc) + pos], cur);
        atomicOr(&mask[0][grp], 1ull << pos);
    }
    __syncthreads();
    uint32_t bufi = 0;
    for (uint32_t l = Lc; l-- > 0;) {  // parent level
        if (emit && live) {
            // the query's ancestor at child level l + 1 and its sibling: a present node (LDS) or the witness entry
            // its parent consumes (same rank as below)
            const CapGroup& d = grp_desc[grp];
            const uint32_t anc = pos >> (Lc - 1u - l), par = anc >> 1;
            const unsigned long long cm = mask[bufi][grp];
            Hash8 sib = zero8();
            if (((cm >> (2 * par)) & 3u) == 3u) sib = load_hash(&xch[bufi][(grp << (l + 1)) + (anc ^ 1u)][0]);
            else {
                const unsigned long long lack = (cm ^ (cm >> 1)) & 0x5555555555555555ull;
                const uint32_t rank = __popcll(lack & ((1ull << (2 * par)) - 1ull));
                const uint32_t base = d.wf ? (uint32_t)d.wf[l + 1] : lvl_s(d.lvl[l + 2]) - d.s_top;
                if (base + rank < d.hw_n) sib = load_hash(d.hw + 8 * (base + rank));
            }
            store_hash(emit + (size_t)(emit_top - l) * 8, sib);
        }
        const uint32_t g2 = t >> l, ppos = t & ((1u << l) - 1u);
        if (g2 < per_block && grp_desc[g2].active) {
            const CapGroup& d = grp_desc[g2];
            const unsigned long long cm = mask[bufi][g2];
            const uint32_t pres = (uint32_t)(cm >> (2 * ppos)) & 3u;
            if (pres) {
                const unsigned long long even = 0x5555555555555555ull;
                const unsigned long long lack = (cm ^ (cm >> 1)) & even;  // bit 2p': exactly one child present
                const uint32_t rank = __popcll(lack & ((1ull << (2 * ppos)) - 1ull));
                const uint32_t base = d.wf ? (uint32_t)d.wf[l + 1] : lvl_s(d.lvl[l + 2]) - d.s_top;
                const uint32_t* kids = &xch[bufi][(g2 << (l + 1)) + 2 * ppos][0];
                Hash8 left, right;
                bool bad = false;
                if (pres == 3u) { left = load_hash(kids); right = load_hash(kids + 8); }
                else {
                    const uint32_t wi = base + rank;
                    Hash8 w8 = zero8();
                    if (wi < d.hw_n) w8 = load_hash(d.hw + 8 * wi);
                    else bad = true;
                    left = (pres & 1u) ? load_hash(kids) : w8;
                    right = (pres & 2u) ? load_hash(kids + 8) : w8;
                }
                Hash8 node = hash_tree(left, right);
                if (l == 0) {
                    if (bad || !hash_eq(node, load_hash(d.root))) atomicOr(d.flags, 1u << d.fail_bit);
                } else {
                    if (bad) atomicOr(d.flags, 1u << d.fail_bit);
                    store_hash(xch[bufi ^ 1][(g2 << l) + ppos], node);
                    atomicOr(&mask[bufi ^ 1][g2], 1ull << ppos);
                }
            }
        }
        __syncthreads();
        if (t < per_block) mask[bufi][t] = 0;
        bufi ^= 1;
        __syncthreads();
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trace_merkle(MerkleArgs a) {
    RSV_TAG(3);
    __shared__ uint32_t xch[2][BLOCK][8];
    __shared__ unsigned long long capmask[2][64];  // per_block <= 64 (the host pads G to >= 4 lanes)
    __shared__ CapGroup capgrp[64];
    const uint32_t G = a.pl.G, per_block = BLOCK / G, Lc = a.Lc;
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    const uint32_t slot_ = blockIdx.x * per_block + grp;
    const int t = blockIdx.y;
    bool live = grp < per_block && slot_ < a.n;
    const uint32_t p = live ? a.pl.proof_of(slot_) : 0u;
    const ProofMeta* m = live ? &a.metas[p] : nullptr;
    live = live && m->reason == R_OK && j < m->nq;
    const uint32_t gbase = grp * G;
    const uint32_t* w = nullptr; const uint32_t* ent = nullptr; const PlanHdr* h = nullptr;
    uint32_t M = 0, A = 0, B = 0, mx = 0, nc_leaf = 0, qv_n = 0, hw_n = 0, s_top = 0, nd_leaf = 0;
    const uint32_t *hw = nullptr, *rows = nullptr;
    bool bad = false;
    Hash8 cur = zero8();
    uint32_t qj = 0;
    if (live) {
        w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
        ent = a.pl.ent + (size_t)slot_ * (a.pl.maxM + 1) * G;
        h = &a.pl.hdr[slot_];
        M = m->M; A = m->A; B = m->B;
        mx = (t == 3) ? M : umax(A, B);
        nc_leaf = (t == 3) ? 8u : ((A == mx ? plonk_cols(t) : 0u) + (B == mx ? poseidon_cols(t) : 0u));
        qv_n = m->qv_n[t];
        hw = w + m->hw_off[t]; hw_n = m->hw_n[t];
        s_top = lvl_s(h->lvl[mx + 1]);
        nd_leaf = lvl_nd(h->lvl[mx]);
        qj = a.ctxs[p].q[j];
        rows = a.rowh + (((size_t)p * 4 + t) * 2) * a.Grow * 8;
        const uint32_t row = ent_rb(ent[mx * G + j]);
        if ((row + 1) * nc_leaf > qv_n) bad = true;
        else {
            cur = load_hash(rows + (size_t)row * 8);
            if (a.path_cols) {
                uint32_t* pc = a.path_cols + (((size_t)slot_ * 4 + t) * G + a.ctxs[p].qperm[j]) * 64;
                const uint32_t* src = w + m->qv_off[t] + row * nc_leaf;
                for (uint32_t k = 0; k < nc_leaf; k++) pc[k] = src[k];
            }
        }
    }
    if (Lc && j == 0 && grp < per_block) {
        CapGroup& d = capgrp[grp];
        d.active = live ? 1u : 0u;
        if (live) {
            d.hw = hw; d.hw_n = hw_n; d.lvl = h->lvl; d.wf = nullptr; d.s_top = s_top;
            d.root = w + W_COMMIT0 + 8 * t; d.flags = &a.ctxs[p].flags; d.fail_bit = R_MERKLE_T0 + t;
        }
    }
    for (uint32_t lvl = a.pl.maxM; lvl > Lc; lvl--) {  // child level
        const uint32_t buf = lvl & 1u;
        const bool on = live && lvl <= mx;
        if (on) store_hash(xch[buf][threadIdx.x], cur);
        __syncthreads();
        if (on) {
            uint32_t e = ent[lvl * G + j];
            Hash8 sib;
            if (ent_sib(e) != 0xFFu) sib = load_hash(xch[buf][gbase + ent_sib(e)]);
            else {
                uint32_t wi = lvl_s(h->lvl[lvl + 1]) - s_top + ent_lb(e);
                if (wi < hw_n) sib = load_hash(hw + 8 * wi);
                else { sib = zero8(); bad = true; }
            }
            bool odd = (qj >> (M - lvl)) & 1u;
            if (a.path_sib) {
                const uint32_t oi = a.ctxs[p].qperm[j];
                store_hash(a.path_sib + ((((size_t)slot_ * 4 + t) * G + oi) * a.pl.maxM + (mx - lvl)) * 8, sib);
                if (lvl == mx) a.path_pos[((size_t)slot_ * 4 + t) * G + oi] = qj >> (M - mx);
            }
            cur = hash_tree_swap(cur, sib, odd);
            const uint32_t pl_ = lvl - 1;  // parent level
            uint32_t nc = (t == 3 || pl_ == mx) ? 0u : ((pl_ == A ? plonk_cols(t) : 0u) + (pl_ == B ? poseidon_cols(t) : 0u));
            if (nc) {
                // lower-level rows were hashed counting from the end of queried_values
                const uint32_t nd_lower = lvl_nd(h->lvl[pl_]), row = ent_rb(ent[pl_ * G + j]);
                const uint32_t off = nd_leaf * nc_leaf + row * nc;
                if (off + nc > qv_n || nd_lower - 1 - row >= a.Grow) bad = true;
                else {
                    cur = combine_with_column(cur, load_hash(rows + ((size_t)a.Grow + (nd_lower - 1 - row)) * 8));
                    if (a.path_cols) {
                        uint32_t* pc = a.path_cols + (((size_t)slot_ * 4 + t) * G + a.ctxs[p].qperm[j]) * 64 + nc_leaf;
                        const uint32_t* src = w + m->qv_off[t] + off;
                        for (uint32_t k = 0; k < nc; k++) pc[k] = src[k];
                    }
                }
            }
        }
    }
    if (live) {
        // every witness hash and queried value must be consumed (components/hints/src/decommit.rs:141-142)
        uint32_t lower = (t == 3 || A == B) ? 0u : umin(A, B);
        uint32_t nc_lower = (t == 3 || A == B) ? 0u : (lower == A ? plonk_cols(t) : poseidon_cols(t));
        uint32_t want_qv = nd_leaf * nc_leaf + (lower ? lvl_nd(h->lvl[lower]) * nc_lower : 0u);
        uint32_t want_hw = lvl_s(h->lvl[1]) - s_top;
        bool ok = !bad && want_qv == qv_n && want_hw == hw_n && (Lc || hash_eq(cur, load_hash(w + W_COMMIT0 + 8 * t)));
        if (!ok) atomicOr(&a.ctxs[p].flags, 1u << (R_MERKLE_T0 + t));
    }
    if (Lc) {
        uint32_t* emit = (a.path_sib && live) ? a.path_sib + (((size_t)slot_ * 4 + t) * G + a.ctxs[p].qperm[j]) * a.pl.maxM * 8 : nullptr;
        merkle_cap<BLOCK>(xch, capmask, capgrp, Lc, per_block, live, grp, live ? (qj >> (M - Lc)) : 0u, cur, emit, mx - 1u);
    }
}

// ------------------------------------------------------------ k_pair_merkle
// SinglePairMerkleProofVar::verify (components/recursive/data_structures/src/lib.rs:400-464):
// blockIdx.y = 0 is the FRI first-layer tree (one QM31 column at each distinct
// column log size), blockIdx.y = 1 + i the i-th inner layer (one column at the leaves).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pair_merkle(MerkleArgs a) {
    RSV_TAG(4);
    __shared__ uint32_t xch[2][BLOCK][8];
    __shared__ uint32_t xch2[BLOCK][8];  // path emission only: pre-column node hashes at data levels
    __shared__ unsigned long long capmask[2][64];  // per_block <= 64 (the host pads G to >= 4 lanes)
    __shared__ CapGroup capgrp[64];
    const uint32_t G = a.pl.G, per_block = BLOCK / G, Lc = a.Lc;
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    const uint32_t slot_ = blockIdx.x * per_block + grp;
    const uint32_t slot = blockIdx.y;
    bool live = grp < per_block && slot_ < a.n;
    const uint32_t p = live ? a.pl.proof_of(slot_) : 0u;
    const ProofMeta* m = live ? &a.metas[p] : nullptr;
    live = live && m->reason == R_OK && j < m->nq && (slot == 0 || slot - 1 < m->n_inner);
    const uint32_t gbase = grp * G;
    const uint32_t* w = nullptr; const uint32_t* ent = nullptr; const PlanHdr* h = nullptr;
    const uint32_t* fl = nullptr; const uint32_t* leafv = nullptr; const ProofCtx* c = nullptr;
    const FriLayerRef* L = nullptr;
    uint32_t M = 0, top = 0, qj = 0, s_top = 0, dslot = 0;
    bool bad = false, have_sib = false;
    uint32_t* psib = nullptr;
    Hash8 cur = zero8(), sibh = zero8();
    if (live) {
        w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
        ent = a.pl.ent + (size_t)slot_ * (a.pl.maxM + 1) * G;
        fl = a.pl.fl + (size_t)slot_ * 2 * G;
        h = &a.pl.hdr[slot_];
        c = &a.ctxs[p];
        M = m->M;
        L = slot == 0 ? &m->first : &m->inner[slot - 1];
        top = slot == 0 ? M : M - slot;  // leaf level of this tree
        qj = c->q[j];
        s_top = lvl_s(h->lvl[top]);
        leafv = a.leafv + ((size_t)slot_ * (3 + a.maxInner)) * G * 8;
        const uint32_t* lv = leafv + ((size_t)(slot == 0 ? 0 : 2 + slot) * G + j) * 8;
        cur = leaf_from_capacity(sponge_capacity4(lv[0], lv[1], lv[2], lv[3]));
        sibh = leaf_from_capacity(sponge_capacity4(lv[4], lv[5], lv[6], lv[7]));
        have_sib = true;
        dslot = 0;
        if (a.pair_sib) {
            const uint32_t oi = c->qperm[j];
            const size_t row = ((size_t)slot_ * (1 + a.maxInner) + slot) * G + oi;
            psib = a.pair_sib + row * a.pl.maxM * 8;
            uint32_t* pc = a.pair_cols + row * 3 * 8;
            const uint32_t nlev = slot == 0 ? c->n_sizes : 1u;
            for (uint32_t g = 0; g < nlev; g++) {
                const uint32_t* src = leafv + ((size_t)(slot == 0 ? g : 2 + slot) * G + j) * 8;
                for (int k = 0; k < 8; k++) pc[g * 8 + k] = src[k];
            }
        }
    }
    if (Lc && j == 0 && grp < per_block) {
        CapGroup& d = capgrp[grp];
        d.active = live ? 1u : 0u;
        if (live) {
            d.hw = w + L->hash_off; d.hw_n = L->hash_n; d.lvl = h->lvl; d.wf = slot == 0 ? h->wf : nullptr; d.s_top = s_top;
            d.root = w + L->commit_off; d.flags = &a.ctxs[p].flags; d.fail_bit = slot == 0 ? R_FRI_FIRST : R_FRI_INNER;
        }
    }
    for (uint32_t lvl = a.pl.maxM; lvl > Lc; lvl--) {  // child level
        const bool on = live && lvl <= top;
        const uint32_t pl_ = lvl - 1;
        // is the parent level a data level of the first-layer tree?
        int dg = -1;
        if (on && slot == 0)
            for (uint32_t g = 1; g < c->n_sizes; g++)
                if (c->sizes[g] == pl_) dg = (int)g;
        // phase A: sibling hash at the child level
        if (on && !have_sib) store_hash(xch[0][threadIdx.x], cur);
        __syncthreads();
        if (on) {
            if (!have_sib) {
                uint32_t e = ent[lvl * G + j];
                if (ent_sib(e) != 0xFFu) sibh = load_hash(xch[0][gbase + ent_sib(e)]);
                else {
                    uint32_t wi;
                    if (slot == 0) wi = dg >= 0 ? (fl[dslot * G + j] & 0xFFFFu) : (uint32_t)h->wf[lvl] + ent_lb(e);
                    else wi = lvl_s(h->lvl[lvl + 1]) - s_top + ent_lb(e);
                    if (wi < L->hash_n) sibh = load_hash(w + L->hash_off + 8 * wi);
                    else { sibh = zero8(); bad = true; }
                }
            }
            bool odd = (qj >> (M - lvl)) & 1u;
            // sibling_hashes[top-1-lvl]: the sibling at a level without a column (data levels: stored in phase B)
            if (psib && !have_sib) store_hash(psib + (size_t)(top - 1 - lvl) * 8, sibh);
            cur = hash_tree_swap(cur, sibh, odd);
            have_sib = false;
        }
        // phase B: data level of the first-layer tree: fold in the column and build the sibling node
        if (on && dg >= 0) {
            const uint32_t* lv = leafv + ((size_t)dg * G + j) * 8;
            if (a.pair_sib) store_hash(xch2[threadIdx.x], cur);  // hash of this node's children, before the column
            cur = combine_with_column(cur, sponge_capacity4(lv[0], lv[1], lv[2], lv[3]));
            store_hash(xch[1][threadIdx.x], cur);
        }
        __syncthreads();
        if (on && dg >= 0) {
            const uint32_t* lv = leafv + ((size_t)dg * G + j) * 8;
            uint32_t w_sib = fl[dslot * G + j] >> 16;
            if (w_sib == 0xFFFFu) {
                uint32_t e = ent[pl_ * G + j];
                if (ent_sib(e) != 0xFFu) {
                    sibh = load_hash(xch[1][gbase + ent_sib(e)]);
                    if (psib) store_hash(psib + (size_t)(top - 1 - pl_) * 8, load_hash(xch2[gbase + ent_sib(e)]));
                } else bad = true;
            } else if (w_sib + 1 < L->hash_n) {
                Hash8 sn = hash_tree(load_hash(w + L->hash_off + 8 * w_sib), load_hash(w + L->hash_off + 8 * (w_sib + 1)));
                if (psib) store_hash(psib + (size_t)(top - 1 - pl_) * 8, sn);
                sibh = combine_with_column(sn, sponge_capacity4(lv[4], lv[5], lv[6], lv[7]));
            } else bad = true;
            have_sib = true;
            dslot++;
        }
    }
    if (live) {
        uint32_t want_hw = slot == 0 ? (uint32_t)h->wf_total : lvl_s(h->lvl[1]) - s_top;
        bool ok = !bad && want_hw == L->hash_n && (Lc || hash_eq(cur, load_hash(w + L->commit_off)));
        if (!ok) atomicOr(&a.ctxs[p].flags, 1u << (slot == 0 ? R_FRI_FIRST : R_FRI_INNER));
    }
    if (Lc) merkle_cap<BLOCK>(xch, capmask, capgrp, Lc, per_block, live, grp, live ? (qj >> (M - Lc)) : 0u, cur, live ? psib : nullptr, top - 2u);
}

// --------------------------------------------------------------- k_finalize
__global__ __launch_bounds__(256) void k_finalize(uint32_t n, const ProofMeta* __restrict__ metas,
                                                  const ProofCtx* __restrict__ ctxs, uint8_t* __restrict__ accept,
                                                  uint8_t* __restrict__ reason) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t r = metas[p].reason;
    if (r == R_OK) {
        uint32_t f = ctxs[p].flags;
        r = f ? (uint32_t)(__ffs((int)f) - 1) : R_OK;
    }
    accept[p] = r == R_OK;
    if (reason) reason[p] = (uint8_t)r;
}

// accept bytes -> little-endian bitmap + popcount (the buffer the multi-GPU host all-gathers)
__global__ __launch_bounds__(256) void k_bitmap(const uint8_t* __restrict__ accept, uint32_t n,
                                                uint32_t* __restrict__ bitmap, unsigned long long* __restrict__ count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = i < n && accept[i];
    unsigned long long mask = __ballot(bit);
    uint32_t lane = threadIdx.x & 63;
    if (lane == 0 && i < n) {
        bitmap[i >> 5] = (uint32_t)mask;
        if ((i >> 5) + 1 < (n + 31) / 32) bitmap[(i >> 5) + 1] = (uint32_t)(mask >> 32);
        if (count && mask) atomicAdd(count, (unsigned long long)__popcll(mask));
    }
}

}  // namespace rsv
