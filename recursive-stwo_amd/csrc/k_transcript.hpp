// k_transcript.hpp — Fiat-Shamir transcript: lane form (k_transcript) and DPP-row form (k_transcript_row).  Part of the pipeline described in verify.hpp.
#pragma once
#include "verify_common.hpp"

namespace rsv {

// ------------------------------------------------------------- k_transcript
// FiatShamirResults::compute (components/recursive/fiat_shamir/src/lib.rs:44-130):
// a strictly sequential chain of channel permutations per proof.
// FLOW: also writes the PoseidonFlow records of the channel operations (layout.hpp; rsv_hints_out::d_flow), including
// the surplus query draws the circuit makes (it draws ceil(n_queries / 4) times where ceil(n_queries / 8) hold every
// query, fiat_shamir/src/lib.rs:119-130), and decides whether this proof's records fit the caller's stride.
// PACE: the permutation instance of the channel (poseidon2.hpp): unpaced while the launch leaves a wave alone on its SIMD
// (up to 49 152 proofs: 32 768 proofs 17.6 -> 17.3 ms), paced beyond (131 072 proofs are two waves per SIMD: 68.4 against
// 69.3 ms unpaced; 65 536: 34.25 against 34.3-34.5).
template <bool FLOW, int PHASE = 0, bool PACE = true>
__global__ __launch_bounds__(64) void k_transcript(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                   uint32_t n, const ProofMeta* __restrict__ metas,
                                                   ProofCtx* __restrict__ ctxs, FlowArgs fa) {
    static_assert(!FLOW || PHASE == 0, "the PoseidonFlow records are written by the unsplit kernel");
    RSV_TAG(1);
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const ProofMeta& m = metas[p];
    if (PHASE == 1) {  // next to the parser, as k_transcript_row<1> below: its first checks, enough to read the fixed-offset part
        const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
        if (o1 < o0 || ((o0 | o1) & 3) || (o1 - o0) > (1ull << 30) || ((o1 - o0) >> 2) < SAMPLES.end + 8) return;
    } else if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    Channel<PACE> ch;
    ch.init();
    if (FLOW) {
        const uint32_t total = flow_total(m.nq, m.n_inner, m.last_n, m.A, m.B, m.M);
        const bool fits = total <= fa.stride;
        c.flow_on = fits ? 1u : 0u;
        if (fa.count) fa.count[p] = fits ? total : 0u;
        if (fits) ch.flow = fa.sink(p);
    }
    Hash8 d;
    // Every field element the transcript absorbs is checked for canonicity here, where it is read (layout.hpp).
    uint32_t over = 0;
    auto ldq_chk = [&](const uint32_t* q) {
        const QM31 v = ldq(q);
        over |= (v.a.a >= P) | (v.a.b >= P) | (v.b.a >= P) | (v.b.b >= P);
        return v;
    };
    if (PHASE != 2) {
    ch.mix(load_hash_chk(w + W_COMMIT0, over));
    ch.mix_one(q_from_m(w[W_LP]));  // statement 0: data_structures/src/lib.rs:52-55 (== m.lp, m.lq once the parser has run)
    ch.mix_one(q_from_m(w[W_LQ]));
    ch.mix(load_hash_chk(w + W_COMMIT0 + 8, over));
    d = ch.draw();  // lookup elements z, alpha: data_structures/src/lib.rs:242-245
    stq(c.z, q_lo(d)); stq(c.alpha, q_hi(d));
    ch.mix_two(ldq_chk(w + W_PLONK_SUM), ldq_chk(w + W_POSEIDON_SUM));  // statement 1: data_structures/src/lib.rs:85-87
    ch.mix(load_hash_chk(w + W_COMMIT0 + 16, over));
    d = ch.draw();
    stq(c.rc, q_lo(d));
    ch.mix(load_hash_chk(w + W_COMMIT0 + 24, over));
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
        ch.mix_two(ldq_chk(w + SAMPLES.off[k]), ldq_chk(w + SAMPLES.off[k + 1]));
    d = ch.draw();
    stq(c.after, q_lo(d));
    if (PHASE == 1) {  // hand the digest (and what the canonicity check found) to the back half
        store_hash(c.pow_digest, ch.digest);
        c.front_over = over;
        return;
    }
    }  // PHASE != 2
    if (PHASE == 2) {
        ch.digest = load_hash(c.pow_digest);
        over = c.front_over;
    }
    ch.mix(load_hash_chk(w + m.first.commit_off, over));
    d = ch.draw();
    stq(c.fri_alpha[0], q_lo(d));
#pragma unroll 1
    for (uint32_t i = 0; i < m.n_inner; i++) {
        ch.mix(load_hash_chk(w + m.inner[i].commit_off, over));
        d = ch.draw();
        stq(c.fri_alpha[i + 1], q_lo(d));
    }
#pragma unroll 1
    for (uint32_t i = 0; i < m.last_n; i += 2) {  // fiat_shamir/src/lib.rs:94-100
        const uint32_t* cf = w + m.last_off + 4 * i;
        if (i + 1 < m.last_n) ch.mix_two(ldq_chk(cf), ldq_chk(cf + 4));
        else ch.mix_one(ldq_chk(cf));
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
    if (FLOW)  // the circuit's surplus draws: same channel, no new value
        for (uint32_t draws = (m.nq + 7u) / 8u; draws < (m.nq + 3u) / 4u; draws++) (void)ch.draw();
    if (over) flags |= 1u << R_PARSE;
    if (flags) atomicOr(&c.flags, flags);  // k_parse zeroed it; k_row_hash may be raising bits beside this kernel
}

// --------------------------------------------------------- k_transcript_row
// The same transcript with ONE PROOF PER 16-LANE ROW (poseidon2_row.hpp): lane i holds state word i, lanes
// 0..7 are the rate half (what is mixed in / drawn), lanes 8..15 the capacity half, i.e. the channel digest
// (primitives/channel/src/lib.rs:30-58).  A permutation is ~4x shorter in latency and ~4x dearer in issue
// slots than in k_transcript, so the host uses this kernel for small batches, where the 233-step chain —
// not throughput — is the cost.
//
// PHASE 0 is the whole transcript.  PHASES 1 and 2 are its two halves as separate launches: the FRONT (the four
// commitments, the statement, the 142 sampled values — everything that sits at a fixed offset of a proof of this type,
// about 85 of the 233 permutations) needs nothing from the parser and runs NEXT TO it on another stream, the BACK (FRI
// commitments, last-layer polynomial, proof of work, queries) needs the section offsets the parser found.  The digest
// travels from one to the other through ProofCtx::pow_digest.  A small batch's step is a chain of dependent kernels,
// and this takes the parser (0.1 ms for 1 024 proofs: ~45 dependent HBM round trips) off that chain.
// FLOW (PHASE 0 only): the row also writes the PoseidonFlow records of the channel operations, as k_transcript<true> does
// — every lane its own word of the input and of the output state (64 + 64 contiguous bytes per record) — so that a small
// batch's flow does not wait for the 3 ms lane-form chain.
template <int PHASE, bool FLOW = false>
__global__ __launch_bounds__(256) void k_transcript_row(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                        uint32_t n, const ProofMeta* __restrict__ metas,
                                                        ProofCtx* __restrict__ ctxs, FlowArgs fa) {
    static_assert(!FLOW || PHASE == 0, "the PoseidonFlow records are written by the unsplit kernel");
    // The step waits for this chain while the row hashes (side stream) share its SIMDs: its waves go first in the arbitration
    // (4 096 proofs 2.65 -> 2.56 ms; nothing at 16 384 and for the lane form above, whose wave is alone with its latency)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t i = threadIdx.x & 15u;
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= n) return;  // whole rows leave together (DPP needs every lane of a live row)
    const ProofMeta& m = metas[p];
    if (PHASE == 1) {
        // the parser has not run yet: its own first checks (k_parse), enough to read the fixed-offset part safely.
        // What the front computes for a proof the parser goes on to reject is never looked at.
        const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
        if (o1 < o0 || ((o0 | o1) & 3) || (o1 - o0) > (1ull << 30) || ((o1 - o0) >> 2) < SAMPLES.end + 8) return;
    } else if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    const bool rate = i < 8;
    const RowRC rc = load_row_rc(i);
    uint32_t dg = 0, n_sent = 0;
    uint32_t* frec = nullptr;  // FLOW: this proof's records, word-addressed (nullptr: none)
    uint32_t fidx = 0;
    if (FLOW) {
        const uint32_t total = flow_total(m.nq, m.n_inner, m.last_n, m.A, m.B, m.M);
        const bool fits = total <= fa.stride;
        if (i == 0) {
            c.flow_on = fits ? 1u : 0u;
            if (fa.count) fa.count[p] = fits ? total : 0u;
        }
        if (fits) frec = fa.rec + (size_t)p * fa.stride * FLOW_WORDS;
    }
    // record of one channel operation: lane i holds word i of the input (left || right = rate || digest half) and of the output
    auto record = [&](uint32_t in, uint32_t out) {
        if (FLOW && frec) {
            frec[(size_t)fidx * FLOW_WORDS + i] = in;
            frec[(size_t)fidx * FLOW_WORDS + 16 + i] = out;
            if (i == 0) fa.swap[(size_t)p * fa.stride + fidx] = 0;
            fidx++;
        }
    };
    // mix: digest = perm(left || digest)[8..16]
    auto mix = [&](uint32_t left_word) {
        const uint32_t in = rate ? left_word : dg;
        uint32_t out = poseidon2_row(in, i, rc);
        record(in, out);
        if (!rate) dg = out;
        n_sent = 0;
    };
    // draw: perm([n_sent, 0 x 7] || digest)[0..8]; the digest is not advanced
    auto draw = [&]() {
        const uint32_t in = rate ? (i == 0 ? n_sent : 0u) : dg;
        uint32_t out = poseidon2_row(in, i, rc);
        record(in, out);
        n_sent++;
        return out;
    };
    // Every word the chain absorbs is fetched BEFORE the permutation that precedes its use: a load issued between
    // two permutations is consumed at once and costs the row its whole HBM latency (1-2 us, against 2-3 us for the
    // permutation itself).
    // Everything word_of / sample_at fetch is a field element the transcript absorbs: checked for canonicity here,
    // where it is read (layout.hpp); the lanes' findings are added up over the row at the end.
    uint32_t over = 0;
    auto word_of = [&](const uint32_t* src, uint32_t n_words) {
        const uint32_t v = (rate && i < n_words) ? src[i] : 0u;
        over |= v >= P;
        return v;
    };
    auto store_felt = [&](uint32_t* dst, uint32_t out) { if (i < 4) dst[i] = out; };
    // where lane i's word of the sample pair (k, k + 1) sits, and the word itself.  The offset comes from a table and the
    // word's address from the offset: fetched in the same round they are two dependent loads in front of a permutation
    // (0.25 us of each of the 71), so the offset is fetched one round before the word (sample_at)
    auto sample_off = [&](int k) { return rate ? (uint32_t)SAMPLES.off[k + (i >> 2)] + (i & 3u) : 0u; };
    auto sample_at = [&](uint32_t off) {
        const uint32_t v = rate ? w[off] : 0u;
        over |= v >= P;
        return v;
    };
    uint32_t nxt = 0, out;
    if (PHASE != 2) {
    const uint32_t c0 = word_of(w + W_COMMIT0, 8), c1 = word_of(w + W_COMMIT0 + 8, 8), c2 = word_of(w + W_COMMIT0 + 16, 8);
    const uint32_t c3 = word_of(w + W_COMMIT0 + 24, 8), sums = word_of(w + W_PLONK_SUM, 8);
    const uint32_t lp = w[W_LP], lq = w[W_LQ];  // == m.lp, m.lq once the parser has run
    nxt = sample_at(sample_off(0));
    uint32_t off_nxt = sample_off(2 < N_SAMPLES ? 2 : 0);
    mix(c0);
    mix(i == 0 ? lp : 0u);  // statement 0: data_structures/src/lib.rs:52-55
    mix(i == 0 ? lq : 0u);
    mix(c1);
    out = draw();  // lookup elements z, alpha
    if (i < 4) c.z[i] = out; else if (i < 8) c.alpha[i - 4] = out;
    mix(sums);  // statement 1: the two total sums are adjacent words 2..10
    mix(c2);
    out = draw();
    store_felt(c.rc, out);
    mix(c3);
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
    for (int k = 0; k < N_SAMPLES; k += 2) {  // fiat_shamir/src/lib.rs:68-75
        const uint32_t cur = nxt;
        if (k + 2 < N_SAMPLES) nxt = sample_at(off_nxt);
        if (k + 4 < N_SAMPLES) off_nxt = sample_off(k + 4);
        mix(cur);
    }
    out = draw();
    store_felt(c.after, out);
    if (PHASE == 1) {  // hand the digest (and what the canonicity check found) to the back half
        const uint32_t ov = sum_row(over);
        if (!rate) c.pow_digest[i - 8] = dg;
        if (i == 0) c.front_over = ov;
        return;
    }
    }  // PHASE != 2
    if (PHASE == 2 && !rate) dg = c.pow_digest[i - 8];
    // The parser's findings in registers: `m` is global memory and the channel's stores (challenges, flow records) may alias
    // it as far as the compiler knows, so m.last_n / m.last_off were re-read in every iteration of the last-layer loop — a
    // dependent load in front of each of its 128 permutations (back half 2.53 us per permutation against the front's 2.04).
    const uint32_t last_n = m.last_n, last_off = m.last_off, n_inner = m.n_inner, nq = m.nq, pow_bits = m.pow_bits, nonce_off = m.nonce_off;
    const uint32_t first_commit = word_of(w + m.first.commit_off, 8);
    // the words of the last-layer polynomial that one mix absorbs (odd tail: second felt = 0)
    auto last_word = [&](uint32_t k) {
        const uint32_t left = 4 * (last_n - k);
        return word_of(w + last_off + 4 * k, left < 8 ? left : 8u);
    };
    nxt = n_inner ? word_of(w + m.inner[0].commit_off, 8) : (last_n ? last_word(0) : 0u);
    mix(first_commit);
    out = draw();
    store_felt(c.fri_alpha[0], out);
#pragma unroll 1
    for (uint32_t l = 0; l < n_inner; l++) {
        const uint32_t cur = nxt;
        nxt = l + 1 < n_inner ? word_of(w + m.inner[l + 1].commit_off, 8) : (last_n ? last_word(0) : 0u);
        mix(cur);
        out = draw();
        store_felt(c.fri_alpha[l + 1], out);
    }
    const uint32_t nonce_lo = w[nonce_off], nonce_hi = w[nonce_off + 1];
#pragma unroll 1
    for (uint32_t k = 0; k < last_n; k += 2) {  // fiat_shamir/src/lib.rs:94-100
        const uint32_t cur = nxt;
        if (k + 2 < last_n) nxt = last_word(k + 2);
        mix(cur);
    }
    // nonce split 22/21/21: data_structures/src/lib.rs:197-213, fiat_shamir/src/lib.rs:102-113
    const uint64_t nonce = (uint64_t)nonce_lo | ((uint64_t)nonce_hi << 32);
    mix(i == 0 ? (uint32_t)(nonce & ((1u << 22) - 1)) : i == 1 ? (uint32_t)((nonce >> 22) & ((1u << 21) - 1))
        : i == 2 ? (uint32_t)((nonce >> 43) & ((1u << 21) - 1)) : 0u);
    if (!rate) c.pow_digest[i - 8] = dg;
    {
        uint32_t ov = sum_row(over);
        if (PHASE == 2) ov += c.front_over;
        const uint32_t fl = ((dg & ((1u << pow_bits) - 1u)) ? (1u << R_POW) : 0u) | (ov ? (1u << R_PARSE) : 0u);  // fiat_shamir/src/lib.rs:115-117
        if (i == 8 && fl) atomicOr(&c.flags, fl);  // k_parse zeroed it; k_row_hash may be raising bits beside this kernel
    }
#pragma unroll 1
    for (uint32_t got = 0; got < nq; got += 8) {  // fiat_shamir/src/lib.rs:119-130
        out = draw();
        if (rate && got + i < nq) c.raw_q[got + i] = out;
    }
    if (FLOW)  // the circuit's surplus draws: same channel, no new value
        for (uint32_t draws = (nq + 7u) / 8u; draws < (nq + 3u) / 4u; draws++) (void)draw();
}

}  // namespace rsv
