// k_parse.hpp — wire-format walk (k_parse) and the canonicity fallback scan (scan_proof_words).  Part of the pipeline described in verify.hpp.
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
    // the library's shape limits (include/rsv.h: RSV_MAX_*; a configuration beyond them never gets here: RSV_E_SIZE)
    if (m.lp < 1 || m.lq < 1 || m.lp > 28 || m.lq > 28 || b < 1 || b > 16 || last > 16 || m.pow_bits > 30) return;
    uint32_t A = m.lp + b, B = m.lq + b, M = umax(m.lp + 1, m.lq + 2) + b;
    if (M > MAX_LOG) return;
    if (A < last + b + 1 || B < last + b + 1) return;
    if (r.w[W_NCOMMIT] != 4 || r.w[W_NCOMMIT + 1] != 0 || r.w[W_NTREES] != 4 || r.w[W_NTREES + 1] != 0) return;
    // constant-shape sampled_values: 4 trees of 50/60/16/8 columns with 1 or 2 mask points.  The 4 + 134 length prefixes of
    // that region sit between the sampled values, which this kernel has no other reason to touch: they are checked where the
    // values are read anyway (k_oods / k_oods_row: EvalCtx::smp; a wrong prefix raises RSV_R_PARSE there) — here the 138
    // strided reads fetched the whole 3.4 KB region a second time (10 KB of HBM traffic per proof for a 1 KB job).
    static_assert(tree_cols(0) == 50 && tree_cols(1) == 60 && tree_cols(2) == 16 && tree_cols(3) == 8, "column counts of the four trees");
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

// ------------------------------------------------------------- slot order on the device
// Inside one launch geometry (one n_queries bucket) the slots are ordered by SHAPE CLASS — the two shape words k_parse
// writes: (n_queries, M, n_inner, lowest level | A, B) — so that the lanes of a wavefront, and nearly every workgroup,
// walk trees of one geometry (the multi-proofs standard configuration mixes three: a wavefront of the Merkle kernels
// that holds several runs every column-level call site once per geometry with most lanes masked; ordering the slots
// was worth 11 % in round 1).  Rounds 1-2 did this on the host: 8 bytes per proof read back, a counting sort, one index
// list uploaded — a host round trip in the middle of every call (the row hashes of a 65 536-proof batch started
// 0.4 ms late because of it, and the call could not return before the parser had finished).  For a batch under ONE
// configuration (one n_queries, so one bucket of known size) three small kernels do it instead:
//   k_classify       find-or-insert every proof's shape words in a 64-entry table (atomicCAS), count the classes
//   k_class_offsets  one wave: order the classes by their shape words, exclusive scan of the counts
//   k_scatter_ids    ids[start[class] + rank] = proof; rank from a wave-aggregated atomic cursor
// Unparsed proofs form a class of their own (their lanes are not live anywhere).  More than 63 distinct shapes in one
// batch (only an adversary builds that) share the last class: still correct, merely mixed.  The order inside a class is
// not deterministic; every proof of a class has the same geometry and every output is addressed by proof, not by slot.
constexpr uint32_t N_CLASSES = 64;
struct ClassTable {
    unsigned long long key[N_CLASSES];  // w0 << 32 | w1, 0 = free; entry N_CLASSES - 1 doubles as the overflow class
    uint32_t count[N_CLASSES];
    uint32_t start[N_CLASSES];
    uint32_t cursor[N_CLASSES];
    uint32_t unparsed;                   // proofs the parser rejected: the last slots
    uint32_t unparsed_cursor;
};

constexpr uint32_t CLASSIFY_PER_BLOCK = 1024;  // proofs per workgroup of k_classify
__global__ __launch_bounds__(256) void k_classify(uint32_t n, const uint32_t* __restrict__ shape, ClassTable* __restrict__ tab,
                                                  uint8_t* __restrict__ cls) {
    // per-class counts go through an LDS histogram: ONE global atomic per class and workgroup (a batch has a handful of
    // classes, so per-wave atomics all hit the same few addresses: 65 536 proofs took 0.28 ms that way, the row hashes
    // waiting behind it)
    __shared__ uint32_t hist[N_CLASSES + 1];
    if (threadIdx.x <= N_CLASSES) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t p0 = blockIdx.x * CLASSIFY_PER_BLOCK, p1 = umin(n, p0 + CLASSIFY_PER_BLOCK);
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) {
        uint32_t c = 0xFFu;  // unparsed
        if (shape[2 * p]) {
            const unsigned long long k = ((unsigned long long)shape[2 * p] << 32) | shape[2 * p + 1];
            // open addressing from a hash of the key; the table is all but empty (a handful of shapes per batch)
            uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 58) % (N_CLASSES - 1);
            c = N_CLASSES - 1;
            for (uint32_t probe = 0; probe < N_CLASSES - 1; probe++, h = (h + 1) % (N_CLASSES - 1)) {
                unsigned long long cur = tab->key[h];
                if (cur == 0) cur = atomicCAS(&tab->key[h], 0ull, k);
                if (cur == 0 || cur == k) { c = h; break; }
            }
        }
        cls[p] = (uint8_t)c;
        atomicAdd(&hist[c == 0xFFu ? N_CLASSES : c], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= N_CLASSES && hist[threadIdx.x]) {
        if (threadIdx.x == N_CLASSES) atomicAdd(&tab->unparsed, hist[threadIdx.x]);
        else atomicAdd(&tab->count[threadIdx.x], hist[threadIdx.x]);
    }
}

__global__ __launch_bounds__(64) void k_class_offsets(ClassTable* __restrict__ tab) {
    // lane i owns table entry i: its start = the counts of all entries whose key sorts before its own (ties cannot
    // happen: keys are distinct; the overflow entry has key 0 = sorts first, which is as good a place as any)
    const uint32_t i = threadIdx.x;
    const unsigned long long mine = tab->key[i];
    uint32_t start = 0;
    for (uint32_t j = 0; j < N_CLASSES; j++) {
        const unsigned long long kj = tab->key[j];
        if (kj < mine || (kj == mine && j < i)) start += tab->count[j];
    }
    tab->start[i] = start;
    tab->cursor[i] = 0;
    if (i == 0) tab->unparsed_cursor = 0;
}

__global__ __launch_bounds__(256) void k_scatter_ids(uint32_t n, const uint8_t* __restrict__ cls, ClassTable* __restrict__ tab,
                                                     uint32_t* __restrict__ ids) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = p < n ? cls[p] : 0xFEu;
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(p < n);
    while (todo) {
        const uint32_t lead = (uint32_t)__ffsll((long long)todo) - 1u;
        const uint32_t cc = (uint32_t)__shfl((int)c, (int)lead);
        const unsigned long long same = __ballot(p < n && c == cc);
        uint32_t base = 0;
        if (lane == lead) {
            const uint32_t k = (uint32_t)__popcll(same);
            // unparsed proofs: behind every class (n - unparsed .. n), from a cursor of their own (start[] is unused for them)
            base = cc == 0xFFu ? n - tab->unparsed + atomicAdd(&tab->unparsed_cursor, k) : tab->start[cc] + atomicAdd(&tab->cursor[cc], k);
        }
        base = (uint32_t)__shfl((int)base, (int)lead);
        if (p < n && c == cc) ids[base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = p;
        todo &= ~same;
    }
}

// ------------------------------------------------------------- canonicity fallback
// (layout.hpp, F_RESCAN): a proof in which some stage found a witness list of the wrong length — already rejected by
// that stage — is read once in full (by k_finalize), so that a non-canonical word in the part of the list nobody consumed
// still gives RSV_R_PARSE, the reason include/rsv.h defines for it.  Every field element of a proof must be a
// canonical M31 word (< P); exempt are the words that are not field elements and that nothing else constrains: the
// two halves of the proof-of-work nonce, and the proof's final word, last_layer_poly.log_size — the reference never
// reads it (it takes the size from coeffs.len(), components/hints/src/folding.rs:573, fiat_shamir.rs:196-200), so any
// u32 there verifies.  Wave-cooperative: all 64 lanes call it for the SAME proof and read it with 16-byte coalesced
// loads; returns non-zero (on every lane) when a non-exempt word is not canonical.
__device__ inline uint32_t scan_proof_words(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                            const ProofMeta& m, uint32_t p, uint32_t lane) {
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
    // eight 1 KB rows of the wave in flight per round: one load per round is consumed at once and costs the wave a memory
    // latency per KB (a 117 KB proof: ~60 us; the bench's tampered proofs whose proof of work then fails are scanned here,
    // and were most of k_finalize's 32 us at 1 024 proofs)
    auto check = [&](uint4 x, uint32_t i) {
        const uint32_t o = (x.x >= P) | ((x.y >= P) << 1) | ((x.z >= P) << 2) | ((x.w >= P) << 3);
        if (o) {
            const uint32_t base = head + 4 * i;
            for (int k = 0; k < 4; k++)
                if (((o >> k) & 1) && !exempt(base + k)) bad = 1;
        }
    };
    uint32_t i = lane;
    for (; i + 7 * 64 < nv; i += 8 * 64) {
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = v[i + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; u++) check(x[u], i + 64 * u);
    }
    for (; i < nv; i += 64) check(v[i], i);
    uint32_t tail = head + 4 * nv;
    if (tail + lane < nw) bad |= (w[tail + lane] >= P) && !exempt(tail + lane);
    return __any(bad) ? 1u : 0u;
}

}  // namespace rsv
