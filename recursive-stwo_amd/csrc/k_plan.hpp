// k_plan.hpp — decommitment plan (k_plan_par, k_plan), transcript export, query-independent quotient constants (k_qconst).  Part of the pipeline described in verify.hpp.
#pragma once
#include "verify_common.hpp"

namespace rsv {

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

// One launch over several n_queries buckets of a mixed batch: every per-query kernel takes up to MAX_FUSED argument
// sets (one per bucket: its own lanes-per-proof G, tables and workspace) and a workgroup finds its set from its block
// index.  A batch of many small buckets is then one launch per stage instead of one per bucket and stage, and the
// buckets' tails fill with each other's workgroups.
constexpr int MAX_FUSED = 16;
template <class Args>
struct Fused {
    uint32_t nb;
    uint32_t lds_proofs;                  // k_plan_par: proofs per workgroup its dynamic LDS tables are sized for
    uint32_t first_block[MAX_FUSED + 1];  // first blockIdx.x of set i; [nb] = grid size
    uint8_t y_of[32];                     // k_pair_merkle: FRI tree of blockIdx.y (host_logic.hpp: pair_layer_order); identity elsewhere
    uint32_t interleave;                  // tree kernels: 1 = the XCD-aware order of RSV_TREE_BLOCK (grid x a multiple of 8), 0 = grid row y = tree
    Args a[MAX_FUSED];
};
// dynamic LDS of k_plan_par for workgroups of up to `proofs` proofs
constexpr size_t plan_lds_bytes(uint32_t proofs) { return (size_t)proofs * (32 * 2 * 8 + 3 * 32 * 4); }
// workgroup-uniform: the argument set of this block and its block index inside that set
// Which (workgroup-of-slots, tree) a workgroup of the tree kernels takes.  Grid (x = workgroups of slots, y = trees) is
// dispatched x-fastest, so with row y = tree all workgroups of tree 0 run first, then tree 1, ...: every tree's pass re-reads the
// proofs' plan tables (1.5 KB per proof and tree: ~1 GB per 65 536-proof step) long after the previous pass left the L2.
// interleave: linear id = (slot-workgroup / 8) * 8 T + tree * 8 + (slot-workgroup % 8) — the T trees of a slot-workgroup are
// dispatched within 8 T consecutive ids and, as workgroup i goes to XCD i mod 8, to the SAME XCD (the L2 is per XCD): the
// tables are fetched once.  (Needs grid x rounded up to a multiple of 8; the surplus workgroups leave at once.)
#define RSV_TREE_BLOCK(FUSED_, BXLIN_, TREE_)                                                            \
    uint32_t BXLIN_ = blockIdx.x, TREE_ = blockIdx.y;                                                    \
    if ((FUSED_).interleave) {                                                                           \
        const uint32_t lin_ = blockIdx.y * gridDim.x + blockIdx.x, r_ = lin_ % (8u * gridDim.y);        \
        TREE_ = r_ >> 3;                                                                                 \
        BXLIN_ = (lin_ / (8u * gridDim.y)) * 8u + (r_ & 7u);                                             \
    }                                                                                                    \
    if (BXLIN_ >= (FUSED_).first_block[(FUSED_).nb]) return
// as RSV_FUSED_SELECT, for a workgroup id that is not blockIdx.x
#define RSV_FUSED_SELECT_AT(FUSED_, ARGS_, BX_, ID_)                                                    \
    uint32_t fused_i_ = 0;                                                                               \
    while (fused_i_ + 1 < (FUSED_).nb && (ID_) >= (FUSED_).first_block[fused_i_ + 1]) fused_i_++;         \
    const auto& ARGS_ = (FUSED_).a[fused_i_];                                                            \
    const uint32_t BX_ = (ID_) - (FUSED_).first_block[fused_i_]
#define RSV_FUSED_SELECT(FUSED_, ARGS_, BX_)                                                            \
    uint32_t fused_i_ = 0;                                                                               \
    while (fused_i_ + 1 < (FUSED_).nb && blockIdx.x >= (FUSED_).first_block[fused_i_ + 1]) fused_i_++;    \
    const auto& ARGS_ = (FUSED_).a[fused_i_];                                                            \
    const uint32_t BX_ = blockIdx.x - (FUSED_).first_block[fused_i_]

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
        if (base != m.first.wit_n) flags |= (1u << R_FRI_FIRST) | F_RESCAN;  // list not read completely: layout.hpp
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

struct PlanArgs {
    uint32_t n;
    PlanPtrs pl;
};
__device__ void qconst_row_body(uint32_t block, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets, uint32_t n,
                                const ProofMeta* __restrict__ metas, ProofCtx* __restrict__ ctxs, bool four);
// QCONST (small batches, chain stream layout): the workgroups behind the plan's also compute the query-independent
// quotient constants in their row form (k_qconst_row's body for qconst_n proofs) — both need only the transcript and
// both precede k_query on the step's chain of dependent kernels, where a launch of its own costs 12-25 us.
template <int BLOCK, bool QCONST = false>
__global__ __launch_bounds__(BLOCK) void k_plan_par(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                    const ProofMeta* __restrict__ metas, ProofCtx* __restrict__ ctxs,
                                                    Fused<PlanArgs> f, uint32_t qconst_n) {
    if (QCONST && blockIdx.x >= f.first_block[f.nb]) {
        qconst_row_body(blockIdx.x - f.first_block[f.nb], blob, offsets, qconst_n & 0x7FFFFFFFu, metas, ctxs, (qconst_n >> 31) != 0u);
        return;
    }
    RSV_FUSED_SELECT(f, pa, bx);
    const uint32_t n = pa.n;
    const PlanPtrs& pl = pa.pl;
    // Per-proof tables in DYNAMIC LDS, sized by the launch for the proofs a workgroup really holds (BLOCK / G: 16 for
    // 16-query proofs): 896 bytes per proof.  As static arrays for the worst case (64 proofs of 4 lanes) they took 59 KB,
    // two workgroups per CU, and the kernel — alone on the chip between the transcript and the trace trees — ran a
    // 65 536-proof batch in eight rounds of ~130 us; now every wave slot is filled (plan_lds_bytes()).
    extern __shared__ unsigned long long plan_dyn[];
    const uint32_t G = pl.G, per_block = BLOCK / G;
    const uint32_t pb = f.lds_proofs;             // proofs per workgroup the launch sized the tables for (>= per_block)
    unsigned long long (*F)[32][2] = reinterpret_cast<unsigned long long (*)[32][2]>(plan_dyn);  // [pb] levels 0..30
    uint32_t (*tl)[32] = reinterpret_cast<uint32_t (*)[32]>(plan_dyn + (size_t)pb * 64);         // per level: nodes | lacking << 8
    uint32_t (*tw)[32] = tl + pb;                 // per node level: witness weight of the first-layer pair tree
    uint32_t (*wsum)[32] = tw + pb;               // wf[l + 1]
    __shared__ uint32_t raw[BLOCK], sq[BLOCK];
    __shared__ uint8_t sp[BLOCK];
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    const uint32_t slot = bx * per_block + grp;
    bool livep = grp < per_block && slot < n;
    const uint32_t p = livep ? pl.proof_of(slot) : 0u;
    const ProofMeta* m = livep ? &metas[p] : nullptr;
    livep = livep && m->reason == R_OK;
    const uint32_t nq = livep ? m->nq : 0u, M = livep ? m->M : 1u, A = livep ? m->A : 0u, B = livep ? m->B : 0u;
    const bool live = livep && j < nq;
    ProofCtx* c = livep ? &ctxs[p] : nullptr;
    const uint32_t gbase = grp * G;
    for (uint32_t i = threadIdx.x; i < pb * 32u * 2u; i += BLOCK) (&F[0][0][0])[i] = 0ull;
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
        if (base != m->first.wit_n) flags |= (1u << R_FRI_FIRST) | F_RESCAN;  // list not read completely: layout.hpp
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
    if (m.reason != R_OK || (c.flags & (1u << R_PARSE))) v = k == 0 ? (uint32_t)R_PARSE : 0u;
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

// ------------------------------------------------------------------ k_qconst_row
// The same constants with ONE PROOF PER 16-LANE ROW, for batches too small to fill the machine: k_qconst's 136-step
// chain of alpha powers and its ~134 column terms are pure latency on the critical path of a small batch
// (transcript -> k_qconst -> k_query -> FRI trees).  Lane i owns the powers after^(16t + i) (start after^i, step
// after^16) and therefore the column terms whose running index is i mod 16; the order in which columns meet powers is
// a compile-time table per kind of size group (QTAB), the lanes' partial sums are added with a row all-reduce.
__host__ __device__ constexpr int sample_index_c(int t, int col, int s) {
    return t == 0 ? S_T0 + col : t == 1 ? S_T1 + col : t == 3 ? S_T3 + col
         : S_T2 + (col < 4 ? col : col < 8 ? 4 + 2 * (col - 4) + s : col < 12 ? 12 + (col - 8) : 16 + 2 * (col - 12) + s);
}
struct QTab {
    uint16_t n[4];               // entries per kind: 0 = the M group (composition columns), 1 = plonk columns only,
    uint8_t si[4][N_APOW];       //                   2 = poseidon columns only, 3 = both (equal log sizes)
    uint8_t batch[4][N_APOW];
};
__host__ __device__ constexpr QTab make_qtab() {
    QTab q{};
    for (int kind = 0; kind < 4; kind++) {
        int k = 0;
        const int n_batches = kind == 0 ? 1 : 2;
        for (int bi = 0; bi < n_batches; bi++)
            for (int t = 0; t < 4; t++) {
                int c0 = 0, c1 = 0;
                if (kind == 0) { if (t != 3) continue; c0 = 0; c1 = 8; }
                else {
                    if (t == 3) continue;
                    c0 = (kind == 1 || kind == 3) ? 0 : (int)plonk_cols(t);
                    c1 = (kind == 2 || kind == 3) ? (int)tree_cols(t) : (int)plonk_cols(t);
                }
                for (int col = c0; col < c1; col++) {
                    const int ns = (int)n_samples_of(t, col);
                    if (bi == 1 && ns != 2) continue;
                    q.si[kind][k] = (uint8_t)sample_index_c(t, col, bi == 1 ? 0 : ns - 1);
                    q.batch[kind][k] = (uint8_t)bi;
                    k++;
                }
            }
        q.n[kind] = (uint16_t)k;
    }
    return q;
}
__constant__ QTab QTAB = make_qtab();
static_assert(make_qtab().n[3] <= N_APOW && make_qtab().n[3] == 134 && make_qtab().n[0] == 8, "alpha powers kept for the quotient numerators");

__global__ __launch_bounds__(256) void k_qconst_row(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                    uint32_t n, const ProofMeta* __restrict__ metas,
                                                    ProofCtx* __restrict__ ctxs) {
    qconst_row_body(blockIdx.x, blob, offsets, n & 0x7FFFFFFFu, metas, ctxs, (n >> 31) != 0u);  // bit 31 of n: four rows per proof
}
__device__ void qconst_row_body(uint32_t block, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets, uint32_t n,
                                const ProofMeta* __restrict__ metas, ProofCtx* __restrict__ ctxs, bool four) {
    // four: FOUR rows per proof — rows 0..2 take one column log size each, row 3 writes the alpha powers.  The three sums
    // and the power table are independent chains of QM31 products; on one row after the other they were 31 us of a single
    // proof's 37 us k_plan_par launch (the plan itself: 17 us).  For batches of a few thousand proofs (the host decides):
    // 1 024 proofs 1.338 -> 1.323 ms, but 16 384 proofs 9.24 -> 9.29 (four times the waves, most of them a quarter full).
    const uint32_t i = threadIdx.x & 15u, row = four ? (threadIdx.x >> 4) & 3u : 0u;
    const uint32_t p = (block * blockDim.x + threadIdx.x) >> (four ? 6 : 4);
    if (p >= n) return;  // whole rows leave together (DPP needs every lane of a live row)
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    const uint32_t M = m.M, A = m.A, B = m.B;
    const uint32_t sizes[3] = {M, A == B ? A : umax(A, B), A == B ? 0u : umin(A, B)};
    const uint32_t n_sizes = A == B ? 2u : 3u;
    auto sel = [](bool cnd, QM31 a, QM31 b) {
        return q_mk(cnd ? a.a.a : b.a.a, cnd ? a.a.b : b.a.b, cnd ? a.b.a : b.b.a, cnd ? a.b.b : b.b.b);
    };
    auto sum16 = [](QM31 x) { return q_mk(sum_row(x.a.a), sum_row(x.a.b), sum_row(x.b.a), sum_row(x.b.b)); };
    // alpha_k = -2u * after^k (data_structures.rs:162-189); lane i: k = i, i + 16, ...
    const QM31 after = ldq(c.after), one = q_one();
    const QM31 a2 = q_mul(after, after), a4 = q_mul(a2, a2), a8 = q_mul(a4, a4), a16 = q_mul(a8, a8);
    QM31 start = sel(i & 1u, after, one);
    start = q_mul(start, sel(i & 2u, a2, one));
    start = q_mul(start, sel(i & 4u, a4, one));
    start = q_mul(start, sel(i & 8u, a8, one));
    start = q_mul(start, q_mk(0, 0, m_neg(2), 0));
    if (!four || row == 3u) {
        QM31 cur = start;
#pragma unroll 1
        for (uint32_t k = i; k < (uint32_t)N_APOW; k += 16u) { stq(c.apow[k], cur); cur = q_mul(cur, a16); }
        if (four) return;
    }
    if (four && row >= n_sizes) return;
    const QM31 ox = ldq(c.oods_x), oy = ldq(c.oods_y);
    for (uint32_t g = four ? row : 0u; g < (four ? row + 1u : n_sizes); g++) {
        const uint32_t l = sizes[g];
        const uint32_t kind = l == M ? 0u : (A == B ? 3u : (l == A ? 1u : 2u));
        // batch 0: OODS point; batch 1: OODS - g_{component log size} (answer/src/lib.rs:62-72)
        const uint32_t comp_log = (l == A) ? m.lp : m.lq;
        CPoint step = {GEN_POW.x[31u - comp_log], GEN_POW.y[31u - comp_log]};  // = cp_gen_mul(1 << (31 - comp_log)); the parser admits 1 <= lp, lq <= 28
        step.y = m_neg(step.y);
        const QM31 sx = q_sub(q_mul_m(ox, step.x), q_mul_m(oy, step.y));
        const QM31 sy = q_add(q_mul_m(ox, step.y), q_mul_m(oy, step.x));
        QM31 sa0 = q_zero(), sb0 = q_zero(), sa1 = q_zero(), sb1 = q_zero(), cur = start;
        const uint32_t cnt = QTAB.n[kind];
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cnt; k0 += 16u) {  // uniform trip count; the last round is partly masked
            const uint32_t k = k0 + i;
            if (k < cnt) {
                const bool b1 = QTAB.batch[kind][k] != 0;
                const QM31 v = ldq(w + SAMPLES.off[QTAB.si[kind][k]]);
                const QM31 py = b1 ? sy : oy;
                // complex_conjugate_line_coeffs_var (data_structures.rs:132-160)
                const QM31 ta = q_mul_c(cur, v.b);
                const QM31 tb = q_mul_c(cur, c_sub(c_mul(v.a, py.b), c_mul(v.b, py.a)));
                sa0 = sel(b1, sa0, q_add(sa0, ta)); sb0 = sel(b1, sb0, q_add(sb0, tb));
                sa1 = sel(b1, q_add(sa1, ta), sa1); sb1 = sel(b1, q_add(sb1, tb), sb1);
            }
            cur = q_mul(cur, a16);
        }
        sa0 = sum16(sa0); sb0 = sum16(sb0); sa1 = sum16(sa1); sb1 = sum16(sb1);
        if (i == 0) {
            const uint32_t n_batches = (l == M) ? 1u : 2u;
            for (uint32_t bi = 0; bi < n_batches; bi++) {
                QBatch& qb = c.batch[g][bi];
                const QM31 px = bi ? sx : ox, py = bi ? sy : oy;
                stq(qb.sa, bi ? sa1 : sa0); stq(qb.sb, bi ? sb1 : sb0);
                qb.prx[0] = px.a.a; qb.prx[1] = px.a.b; qb.pix[0] = px.b.a; qb.pix[1] = px.b.b;
                qb.pry[0] = py.a.a; qb.pry[1] = py.a.b; qb.piy[0] = py.b.a; qb.piy[1] = py.b.b;
            }
            c.n_batches[g] = n_batches;
        }
    }
}

}  // namespace rsv
