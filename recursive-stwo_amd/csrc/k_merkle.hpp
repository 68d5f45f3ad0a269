// k_merkle.hpp — column hashing (k_row_hash), trace trees (k_trace_merkle), top-of-tree cap, FRI pair trees (k_pair_merkle).  Part of the pipeline described in verify.hpp.
#pragma once
#include "k_query.hpp"

// waves per SIMD the Merkle kernels are compiled for (register budget 512 / waves, in granules of 8)
#ifndef RSV_PAIR_WAVES
#define RSV_PAIR_WAVES 5
#endif
#ifndef RSV_TRACE_WAVES
#define RSV_TRACE_WAVES 6
#endif

namespace rsv {

// --------------------------------------------------------------- k_row_hash
// Column hashing of the trace trees does not depend on the transcript: queried_values lists one row of
// column values per distinct queried node, in ascending node order, leaf level first.  The sponge over row r
// (primitives/merkle/src/lib.rs:50-181) can therefore run UNDERNEATH the latency-bound transcript kernel
// (side stream): one lane per (proof, tree, row) hashes leaf-level row r counted from the start of
// queried_values and lower-level row r counted from its END (where that block begins depends on how many
// leaves are distinct, which is only known once the queries are).  k_trace_merkle then just picks its rows.
//   rowh[((slot*4 + t)*2 + 0)*G + r] = leaf hash of leaf row r
//   rowh[((slot*4 + t)*2 + 1)*G + r] = column capacity digest of the r-th LAST lower-level row
struct RowHashArgs {
    const uint8_t* blob;
    const uint64_t* offsets;
    uint32_t n, G;         // slots in this launch, lanes per proof (= n_queries of the bucket)
    const ProofMeta* metas;
    uint32_t* rowh;        // [slot][4][2][G][8] (this bucket's part of the row-hash workspace)
    const uint32_t* ids;   // slot -> proof (nullptr: identity)
    ProofCtx* ctxs;        // flags: a non-canonical queried value -> RSV_R_PARSE
};

template <int PACE = 1>
__global__ __launch_bounds__(256) void k_row_hash(Fused<RowHashArgs> f) {
    RSV_TAG(2);
    RSV_FUSED_SELECT(f, a, bx);
    // rows are independent: the lanes are dealt over (slot, row) without regard to workgroup boundaries, so a query count
    // that does not divide 256 (80, 27, 11, 10) leaves no lane idle but the launch's last ones
    const uint32_t G = a.G;
    const uint64_t item = (uint64_t)bx * 256 + threadIdx.x;
    const uint32_t slot = (uint32_t)(item / G), r = (uint32_t)(item % G);
    const int t = blockIdx.y;
    if (slot >= a.n) return;
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
    uint32_t* out = a.rowh + (((size_t)slot * 4 + t) * 2) * G * 8;
    uint32_t over = 0;  // a non-canonical queried value (this kernel is where queried_values are read: layout.hpp)
    if ((r + 1) * nc_leaf <= qv_n)
        store_hash(out + (size_t)r * 8, leaf_from_capacity<PACE>(sponge_capacity_chk<PACE>(qv + r * nc_leaf, nc_leaf, over)));
    if (nc_lower && (r + 1) * nc_lower <= qv_n)
        store_hash(out + ((size_t)G + r) * 8, sponge_capacity_chk<PACE>(qv + qv_n - (r + 1) * nc_lower, nc_lower, over));
    {
        // The words no row covers: empty for every list the Merkle stage accepts (its rows tile the list: nd_leaf <= nq
        // leaf rows from the start, nd_lower <= nq lower rows from the end); a longer or ragged list is rejected there,
        // but its words are field elements all the same and a non-canonical one outranks that rejection.
        const uint32_t nq = m.nq;
        const uint32_t r1 = umin(nq, qv_n / nc_leaf), r2 = nc_lower ? umin(nq, qv_n / nc_lower) : 0u;
        const uint32_t lo = r1 * nc_leaf, hi = qv_n - r2 * nc_lower;
        for (uint32_t i = lo + r; i < hi; i += nq) over |= qv[i] >= P;
    }
    if (over) atomicOr(&a.ctxs[p].flags, 1u << R_PARSE);
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
    const uint32_t* rowh;  // k_row_hash output for the slots of this launch, [slot][4][2][G][8]
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
    // top of the cap in kernels of its own: the in-kernel cap stops at level Lt (0 = it walks to the root; Lt = Lc: it only
    // hands the level-Lc nodes over) and leaves, per (slot, tree), the present nodes of that level and their presence mask:
    //   capn[((slot * T + tree) << Lt) + pos][8], capm[slot * T + tree]    T = 4 (trace trees: t*) / 1 + maxInner (FRI trees: p*)
    // Lt2 < Lt: k_cap_mid walks from level Lt to level Lt2 (a subtree per lane) and leaves capn2 / capm2 in the same layout for
    // k_cap_top; Lt2 == Lt: k_cap_top reads capn / capm itself (host_logic.hpp: cap_levels).
    uint32_t Lt, Lt2;
    uint32_t *tcapn, *pcapn;
    unsigned long long *tcapm, *pcapm;
    uint32_t *tcapn2, *pcapn2;
    unsigned long long *tcapm2, *pcapm2;
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
    uint4* frec;           // FLOW: this proof's PoseidonFlow records (nullptr: none)
    uint8_t* fswap;
};

// emit (optional, path emission): this lane's path buffer; the sibling consumed at child level l + 1 goes to entry
// emit_top - l (8 words each), so a query's path is complete although its lane does not walk the top levels.
// FLOW: a node several queries of a proof share is still ONE permutation here; the lane that hashes it writes the record
// of every query whose path passes through it (the records differ only in which child is "self": swap = the position's
// parity).  fl[0][lane] = record index of that query's step from level Lc (0xFFFFFFFF: none), fl[1][lane] = its position
// at level Lc; the query lanes of group g are threads g * G .. g * G + G - 1.
// BLOCK: the workgroup's lanes as the kernel indexes them (virtual lanes in the row form)
template <int BLOCK, bool FLOW = false, int PACE = 1>
__device__ __forceinline__ void merkle_cap(uint32_t (*xch)[BLOCK][8], unsigned long long (*mask)[64], CapGroup* grp_desc,
                                           uint32_t Lc, uint32_t per_block, bool live, uint32_t grp, uint32_t pos,
                                           const Hash8& cur, uint32_t* emit = nullptr, uint32_t emit_top = 0, uint32_t Lt = 0,
                                           uint32_t* capn = nullptr, unsigned long long* capm = nullptr, uint32_t slot0 = 0, uint32_t n_slots = 0,
                                           uint32_t T = 0, uint32_t ti = 0, const uint32_t (*fl)[FLOW ? BLOCK : 1] = nullptr, uint32_t G = 0) {
    const uint32_t t = vlane<PACE>();
    __syncthreads();  // xch is free, descriptors written
    if (t < per_block) { mask[0][t] = 0; mask[1][t] = 0; }
    __syncthreads();
    if (live) {
        store_hash(xch[0][(grp << Lc) + pos], cur);
        atomicOr(&mask[0][grp], 1ull << pos);
    }
    __syncthreads();
    uint32_t bufi = 0;
    for (uint32_t l = Lc; l-- > Lt;) {  // parent level
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
                    if (wi < d.hw_n) {
                        w8 = load_hash(d.hw + 8 * wi);
                        if (hash_over(w8)) atomicOr(d.flags, 1u << R_PARSE);
                    } else bad = true;
                    left = (pres & 1u) ? load_hash(kids) : w8;
                    right = (pres & 2u) ? load_hash(kids + 8) : w8;
                }
                Hash8 node;
                if constexpr (FLOW) {
                    const State16 st = poseidon2_full<PACE>(join(left, right));
                    node = rate_of(st);
                    if (d.frec) {
                        const FlowSink fs{d.frec, d.fswap};
                        for (uint32_t q = 0; q < G; q++) {
                            const uint32_t idx0 = fl[0][g2 * G + q];
                            if (idx0 == 0xFFFFFFFFu) continue;
                            const uint32_t anc = fl[1][g2 * G + q] >> (Lc - 1u - l);  // the query's ancestor at the child level
                            if ((anc >> 1) != ppos) continue;
                            if (anc & 1u) flow_put(fs, idx0 + (Lc - 1u - l), right, left, st, 1u);
                            else flow_put(fs, idx0 + (Lc - 1u - l), left, right, st, 0u);
                        }
                    }
                } else node = hash_tree<PACE>(left, right);
                if (l == 0) {
                    if (bad || !hash_eq(node, load_hash(d.root))) atomicOr(d.flags, (1u << d.fail_bit) | (bad ? F_RESCAN : 0u));
                } else {
                    if (bad) atomicOr(d.flags, (1u << d.fail_bit) | F_RESCAN);
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
    if (Lt && ti < T) {  // the levels above Lt belong to the cap kernels: hand over this level's nodes (xch[bufi], mask[bufi]);
                         // ti >= T: a launch's grid covers the deepest bucket's FRI layers, this bucket has fewer trees
        const uint32_t g2 = t >> Lt, pp = t & ((1u << Lt) - 1u);
        if (g2 < per_block && slot0 + g2 < n_slots) {
            const unsigned long long cm = grp_desc[g2].active ? mask[bufi][g2] : 0ull;  // (Lt == Lc <= 6: up to 64 positions)
            const size_t idx = (size_t)(slot0 + g2) * T + ti;
            if (pp == 0) capm[idx] = cm;
            if ((cm >> pp) & 1u) store_hash(capn + ((idx << Lt) + pp) * 8, load_hash(&xch[bufi][(g2 << Lt) + pp][0]));
        }
    }
}

// FLOW (a second instantiation, launched only when rsv_hints_out::d_flow is given, always with Lc = 0): every lane also
// writes the PoseidonFlow records of ITS path — leaf sponge, one swap-permute per level, the lower column level's
// sponge and combine — at the index the circuit's invocation order gives them (layout.hpp).  The circuit hashes every
// query's leaf and column rows itself, so in this mode the lane does too (k_row_hash hashes each distinct row once).
// PACE = FORM_ROW (poseidon2.hpp): the same kernel on VIRTUAL lanes — a workgroup of BLOCK threads is BLOCK / 16 lanes of the
// indexing below, each a DPP row whose 16 threads compute the same indices, load the same words and share every
// permutation (poseidon2_row_half).  For launches of a few waves (a single proof: 64 paths per tree), where the walk is
// a chain of dependent permutations and the row form's chain is a quarter as long: one proof's trace trees 285 -> 67 us, its call 1.15 -> 0.82 ms; 128 proofs 1.14 -> 0.91 ms; 512: slower (1.26 -> 1.54).
// Stores and atomicOr's of the 16 threads coincide (same address, same value).
template <int BLOCK, bool FLOW = false, int PACE = 1>
__global__ __launch_bounds__(BLOCK, PACE == FORM_ROW ? 2 : (FLOW ? 4 : RSV_TRACE_WAVES)) void k_trace_merkle(Fused<MerkleArgs> f, FlowArgs fa) {
    RSV_TAG(3);
    RSV_TREE_BLOCK(f, bxlin, tree_y);
    RSV_FUSED_SELECT_AT(f, a, bx, bxlin);
    constexpr int VB = PACE == FORM_ROW ? BLOCK / 16 : BLOCK;  // lanes of this kernel's indexing per workgroup
    const uint32_t tid = vlane<PACE>();
    row_rc_init<PACE>();
    __shared__ uint32_t xch[2][VB][8];
    __shared__ unsigned long long capmask[2][64];  // per_block <= 64 (the host pads G to >= 4 lanes)
    __shared__ CapGroup capgrp[64];
    __shared__ uint32_t xneed;
    __shared__ uint32_t capfl[2][FLOW ? VB : 1];  // FLOW with a cap: see merkle_cap
    const uint32_t G = a.pl.G, per_block = VB / G, Lc = a.Lc;
    const uint32_t grp = tid / G, j = tid % G;
    const uint32_t slot_ = bx * per_block + grp;
    const int t = (int)tree_y;
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
    FlowSink fs{nullptr, nullptr};  // FLOW: this proof's records (rec == nullptr: none)
    uint32_t fbase = 0;             // FLOW: index of this path's first record
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
        rows = a.rowh + (((size_t)slot_ * 4 + t) * 2) * G * 8;
        const uint32_t row = ent_rb(ent[mx * G + j]);
        if (FLOW && a.ctxs[p].flow_on) {
            fs = fa.sink(p);
            fbase = flow_trace_base(t, m->nq, m->n_inner, m->last_n, A, B, M) + a.ctxs[p].qperm[j] * flow_trace_path_len(t, A, B, M);
        }
        if ((row + 1) * nc_leaf > qv_n) bad = true;
        else {
            if (FLOW && fs.rec) {  // hash_m31_columns_get_rate (merkle/src/lib.rs:50-91): the chunks, then the rate permutation
                const Hash8 d = flow_sponge_capacity<PACE>(fs, fbase, w + m->qv_off[t] + row * nc_leaf, nc_leaf);
                cur = rate_of(flow_perm<PACE>(fs, fbase + flow_chunks(nc_leaf), zero8(), d, false));
            } else
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
            d.frec = fs.rec; d.fswap = fs.swap;
        }
    }
    // Sibling exchange through LDS (and its workgroup barrier) only at the levels where some proof of this workgroup
    // has two query paths that are siblings of each other: a level-l node has its sibling on a query path iff the
    // number of distinct nodes grows from level l-1 to l.  Deep in the tree that is rare (16 proofs x 16 queries:
    // ~1 workgroup in 10 at level 14), and a barrier per level makes every wave wait for the slowest of its four.
    if (tid == 0) xneed = 0;
    __syncthreads();
    if (live && j == 0) {
        uint32_t mask = 0;
        for (uint32_t l = Lc + 1; l <= mx; l++)
            if (lvl_nd(h->lvl[l - 1]) < lvl_nd(h->lvl[l])) mask |= 1u << l;
        if (mask) atomicOr(&xneed, mask);
    }
    __syncthreads();
    const uint32_t need = xneed;
    uint32_t buf = 0;  // toggles per exchange (not per level): a buffer is rewritten two barriers after its last read
    for (uint32_t lvl = a.pl.maxM; lvl > Lc; lvl--) {  // child level
        const bool on = live && lvl <= mx;
        const bool exch = (need >> lvl) & 1u;  // workgroup-uniform
        if (exch) {
            if (on) store_hash(xch[buf][tid], cur);
            __syncthreads();
        }
        if (on) {
            uint32_t e = ent[lvl * G + j];
            Hash8 sib;
            if (exch && ent_sib(e) != 0xFFu) sib = load_hash(xch[buf][gbase + ent_sib(e)]);
            else {
                uint32_t wi = lvl_s(h->lvl[lvl + 1]) - s_top + ent_lb(e);
                if (wi < hw_n) {
                    sib = load_hash(hw + 8 * wi);
                    // hash witnesses are read here, so here they are checked for canonicity (layout.hpp); raised at
                    // once (a branch that is almost never taken) rather than carried in a register across the walk
                    if (hash_over(sib)) atomicOr(&a.ctxs[p].flags, 1u << R_PARSE);
                } else { sib = zero8(); bad = true; }
            }
            bool odd = (qj >> (M - lvl)) & 1u;
            if (a.path_sib) {
                const uint32_t oi = a.ctxs[p].qperm[j];
                store_hash(a.path_sib + ((((size_t)slot_ * 4 + t) * G + oi) * a.pl.maxM + (mx - lvl)) * 8, sib);
                if (lvl == mx) a.path_pos[((size_t)slot_ * 4 + t) * G + oi] = qj >> (M - mx);
            }
            const uint32_t pl_ = lvl - 1;  // parent level
            uint32_t nc = (t == 3 || pl_ == mx) ? 0u : ((pl_ == A ? plonk_cols(t) : 0u) + (pl_ == B ? poseidon_cols(t) : 0u));
            // FLOW: records of this step (the circuit's loop index is mx - lvl): [column sponge chunks] swap-permute
            // [combine]; the steps above the lower column level sit behind that level's chunks + combine
            uint32_t fidx = 0;
            if (FLOW && fs.rec) {
                const uint32_t lower = (t == 3 || A == B) ? 0u : umin(A, B);
                const uint32_t extra = (lower && pl_ < lower) ? flow_chunks(lower == A ? plonk_cols(t) : poseidon_cols(t)) + 1u : 0u;
                fidx = fbase + flow_chunks(nc_leaf) + 1u + (mx - lvl) + extra;
                cur = rate_of(flow_perm<PACE>(fs, fidx + (nc ? flow_chunks(nc) : 0u), cur, sib, odd));
            } else
            cur = hash_tree_swap<PACE>(cur, sib, odd);
            if (nc) {
                // lower-level rows were hashed counting from the end of queried_values
                const uint32_t nd_lower = lvl_nd(h->lvl[pl_]), row = ent_rb(ent[pl_ * G + j]);
                const uint32_t off = nd_leaf * nc_leaf + row * nc;
                if (off + nc > qv_n || nd_lower - 1 - row >= G) bad = true;
                else {
                    if (FLOW && fs.rec) {  // hash_tree_with_column_hash_with_swap (merkle/src/lib.rs:32-41)
                        const Hash8 colcap = flow_sponge_capacity<PACE>(fs, fidx, w + m->qv_off[t] + off, nc);
                        cur = rate_of(flow_perm<PACE>(fs, fidx + flow_chunks(nc) + 1u, cur, colcap, false));
                    } else
                    cur = combine_with_column<PACE>(cur, load_hash(rows + ((size_t)G + (nd_lower - 1 - row)) * 8));
                    if (a.path_cols) {
                        uint32_t* pc = a.path_cols + (((size_t)slot_ * 4 + t) * G + a.ctxs[p].qperm[j]) * 64 + nc_leaf;
                        const uint32_t* src = w + m->qv_off[t] + off;
                        for (uint32_t k = 0; k < nc; k++) pc[k] = src[k];
                    }
                }
            }
        }
        if (exch) buf ^= 1u;
    }
    if (live) {
        // every witness hash and queried value must be consumed (components/hints/src/decommit.rs:141-142)
        uint32_t lower = (t == 3 || A == B) ? 0u : umin(A, B);
        uint32_t nc_lower = (t == 3 || A == B) ? 0u : (lower == A ? plonk_cols(t) : poseidon_cols(t));
        uint32_t want_qv = nd_leaf * nc_leaf + (lower ? lvl_nd(h->lvl[lower]) * nc_lower : 0u);
        uint32_t want_hw = lvl_s(h->lvl[1]) - s_top;
        bool ok = !bad && want_qv == qv_n && want_hw == hw_n && (Lc || hash_eq(cur, load_hash(w + W_COMMIT0 + 8 * t)));
        // a witness list of the wrong length is not read completely: k_rescan reads the whole proof (layout.hpp)
        const uint32_t fl = (ok ? 0u : 1u << (R_MERKLE_T0 + t)) | ((bad || want_hw != hw_n) ? F_RESCAN : 0u);
        if (fl) atomicOr(&a.ctxs[p].flags, fl);
    }
    if (Lc) {
        uint32_t* emit = (a.path_sib && live) ? a.path_sib + (((size_t)slot_ * 4 + t) * G + a.ctxs[p].qperm[j]) * a.pl.maxM * 8 : nullptr;
        if constexpr (FLOW) {
            // this path's record of the step from level Lc: behind the leaf sponge, the steps above and — every column
            // level lies above the cap — the lower column level's chunks + combine
            const uint32_t lower = (live && t != 3 && A != B) ? umin(A, B) : 0u;
            const uint32_t extra = lower ? flow_chunks(lower == A ? plonk_cols(t) : poseidon_cols(t)) + 1u : 0u;
            capfl[0][tid] = (live && fs.rec) ? fbase + flow_chunks(nc_leaf) + 1u + (mx - Lc) + extra : 0xFFFFFFFFu;
            capfl[1][tid] = live ? (qj >> (M - Lc)) : 0u;
        }
        merkle_cap<VB, FLOW, PACE>(xch, capmask, capgrp, Lc, per_block, live, grp, live ? (qj >> (M - Lc)) : 0u, cur, emit, mx - 1u, a.Lt, a.tcapn,
                                a.tcapm, bx * per_block, a.n, 4u, (uint32_t)t, capfl, G);
    }
}

// ------------------------------------------------------------ k_pair_merkle
// SinglePairMerkleProofVar::verify (components/recursive/data_structures/src/lib.rs:400-464):
// blockIdx.y = 0 is the FRI first-layer tree (one QM31 column at each distinct
// column log size), blockIdx.y = 1 + i the i-th inner layer (one column at the leaves).
// FLOW (second instantiation, Lc = 0, with the dynamic LDS buffer): every lane also writes the PoseidonFlow records of
// its pair path (layout.hpp): the two leaf sponges, one swap-permute per level and, at a lower column level of the
// first-layer tree, the two column capacities and the two combines — the sibling node's combine is then computed by
// the lane itself from the sibling's children hash (another lane's, through xch2, or hashed from the witness pair,
// which the circuit takes as a hint and does not hash).
// PACE = FORM_ROW: on virtual lanes, as k_trace_merkle (one proof's FRI trees 435 -> 94 us).
template <int BLOCK, bool FLOW = false, int PACE = 1>
__global__ __launch_bounds__(BLOCK, PACE == FORM_ROW ? 2 : (FLOW ? 4 : RSV_PAIR_WAVES)) void k_pair_merkle(Fused<MerkleArgs> f, FlowArgs fa) {
    RSV_TAG(4);
    RSV_TREE_BLOCK(f, bxlin, tree_y);
    RSV_FUSED_SELECT_AT(f, a, bx, bxlin);
    constexpr int VB = PACE == FORM_ROW ? BLOCK / 16 : BLOCK;  // lanes of this kernel's indexing per workgroup
    const uint32_t tid = vlane<PACE>();
    row_rc_init<PACE>();
    __shared__ uint32_t xch[2][VB][8];   // phase A (sibling hashes), toggled per exchange; reused by merkle_cap
    __shared__ uint32_t xcol[VB][8];     // phase B (nodes with their column folded in), data levels only
    // path emission only: pre-column node hashes at data levels.  Dynamic LDS (BLOCK x 32 bytes, passed by the launch
    // only when pair paths are emitted): without it the kernel holds 29 KB of LDS and five workgroups fit a CU.
    extern __shared__ uint32_t xch2_dyn[];
    uint32_t (*xch2)[8] = reinterpret_cast<uint32_t (*)[8]>(xch2_dyn);
    __shared__ unsigned long long capmask[2][64];  // per_block <= 64 (the host pads G to >= 4 lanes)
    __shared__ CapGroup capgrp[64];
    __shared__ uint32_t xneed[2];
    __shared__ uint32_t capfl[2][FLOW ? VB : 1];  // FLOW with a cap: see merkle_cap
    const uint32_t G = a.pl.G, per_block = VB / G, Lc = a.Lc;
    const uint32_t grp = tid / G, j = tid % G;
    const uint32_t slot_ = bx * per_block + grp;
    const uint32_t slot = f.y_of[tree_y];  // which FRI tree this workgroup hashes (a permutation chosen by the host for small launches)
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
    FlowSink fs{nullptr, nullptr};  // FLOW: this proof's records (rec == nullptr: none)
    uint32_t fbase = 0;             // FLOW: index of this path's first record
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
        if (FLOW && c->flow_on) {
            fs = fa.sink(p);
            fbase = flow_pair_base(slot, m->nq, m->n_inner, m->last_n, m->A, m->B, M) + c->qperm[j] * flow_pair_path_len(slot, m->A, m->B, M);
        }
        if (FLOW && fs.rec) {  // hash_qm31_columns_get_rate(&[v, 0]) of the query's own value, then of its pair sibling's
            cur = rate_of(flow_perm<PACE>(fs, fbase + 1u, zero8(), flow_capacity4<PACE>(fs, fbase, lv), false));
            sibh = rate_of(flow_perm<PACE>(fs, fbase + 3u, zero8(), flow_capacity4<PACE>(fs, fbase + 2u, lv + 4), false));
        } else {
        cur = leaf_from_capacity<PACE>(sponge_capacity4<PACE>(lv[0], lv[1], lv[2], lv[3]));
        sibh = leaf_from_capacity<PACE>(sponge_capacity4<PACE>(lv[4], lv[5], lv[6], lv[7]));
        }
        have_sib = true;
        dslot = 0;
        if (a.pair_sib || a.pair_cols) {  // (the column values may be asked for without the sibling paths)
            const uint32_t oi = c->qperm[j];
            const size_t row = ((size_t)slot_ * (1 + a.maxInner) + slot) * G + oi;
            if (a.pair_sib) psib = a.pair_sib + row * a.pl.maxM * 8;
            if (a.pair_cols) {
                uint32_t* pc = a.pair_cols + row * 3 * 8;
                const uint32_t nlev = slot == 0 ? c->n_sizes : 1u;
                for (uint32_t g = 0; g < nlev; g++) {
                    const uint32_t* src = leafv + ((size_t)(slot == 0 ? g : 2 + slot) * G + j) * 8;
                    for (int k = 0; k < 8; k++) pc[g * 8 + k] = src[k];
                }
            }
        }
    }
    if (Lc && j == 0 && grp < per_block) {
        CapGroup& d = capgrp[grp];
        d.active = live ? 1u : 0u;
        if (live) {
            d.hw = w + L->hash_off; d.hw_n = L->hash_n; d.lvl = h->lvl; d.wf = slot == 0 ? h->wf : nullptr; d.s_top = s_top;
            d.root = w + L->commit_off; d.flags = &a.ctxs[p].flags; d.fail_bit = slot == 0 ? R_FRI_FIRST : R_FRI_INNER;
            d.frec = fs.rec; d.fswap = fs.swap;
        }
    }
    // Workgroup barriers only where lanes really exchange through LDS (see k_trace_merkle): [0] child levels at which
    // some proof of this workgroup has sibling query paths, [1] child levels whose parent carries a column of the
    // first-layer tree.  The column levels also take the phase-A barrier, which orders the reuse of xcol / xch2.
    if (tid < 2) xneed[tid] = 0;
    __syncthreads();
    if (live && j == 0) {
        uint32_t ma = 0, mb = 0;
        for (uint32_t l = Lc + 1; l <= top; l++)
            if (lvl_nd(h->lvl[l - 1]) < lvl_nd(h->lvl[l])) ma |= 1u << l;
        if (slot == 0)
            for (uint32_t g = 1; g < c->n_sizes; g++)
                if (c->sizes[g] < top) mb |= 1u << (c->sizes[g] + 1u);
        if (ma | mb) atomicOr(&xneed[0], ma | mb);
        if (mb) atomicOr(&xneed[1], mb);
    }
    __syncthreads();
    const uint32_t needA = xneed[0], needB = xneed[1];
    uint32_t buf = 0;
    for (uint32_t lvl = a.pl.maxM; lvl > Lc; lvl--) {  // child level
        const bool on = live && lvl <= top;
        const uint32_t pl_ = lvl - 1;
        uint32_t fidx = 0;
        const bool exA = (needA >> lvl) & 1u, exB = (needB >> lvl) & 1u;  // workgroup-uniform
        // is the parent level a data level of the first-layer tree?
        int dg = -1;
        if (on && exB)
            for (uint32_t g = 1; g < c->n_sizes; g++)
                if (c->sizes[g] == pl_) dg = (int)g;
        // phase A: sibling hash at the child level
        if (exA) {
            if (on && !have_sib) store_hash(xch[buf][tid], cur);
            __syncthreads();
        }
        if (on) {
            if (!have_sib) {
                uint32_t e = ent[lvl * G + j];
                if (exA && ent_sib(e) != 0xFFu) sibh = load_hash(xch[buf][gbase + ent_sib(e)]);
                else {
                    uint32_t wi;
                    if (slot == 0) wi = dg >= 0 ? (fl[dslot * G + j] & 0xFFFFu) : (uint32_t)h->wf[lvl] + ent_lb(e);
                    else wi = lvl_s(h->lvl[lvl + 1]) - s_top + ent_lb(e);
                    if (wi < L->hash_n) {
                        sibh = load_hash(w + L->hash_off + 8 * wi);
                        if (hash_over(sibh)) atomicOr(&a.ctxs[p].flags, 1u << R_PARSE);  // as in k_trace_merkle
                    } else { sibh = zero8(); bad = true; }
                }
            }
            bool odd = (qj >> (M - lvl)) & 1u;
            // sibling_hashes[top-1-lvl]: the sibling at a level without a column (data levels: stored in phase B)
            if (psib && !have_sib) store_hash(psib + (size_t)(top - 1 - lvl) * 8, sibh);
            // FLOW: the records of this step (circuit loop index top - lvl, behind four extra records per column level
            // already passed): swap-permute alone, or [self column capacity, sibling column capacity, swap-permute,
            // combine self, combine sibling] when the parent level carries a column
            fidx = fbase + 4u + (top - lvl) + 4u * dslot;
            if (FLOW && fs.rec) cur = rate_of(flow_perm<PACE>(fs, fidx + (dg >= 0 ? 2u : 0u), cur, sibh, odd));
            else
            cur = hash_tree_swap<PACE>(cur, sibh, odd);
            have_sib = false;
        }
        // phase B: data level of the first-layer tree: fold in the column and build the sibling node
        if (on && dg >= 0) {
            const uint32_t* lv = leafv + ((size_t)dg * G + j) * 8;
            if (a.pair_sib || FLOW) store_hash(xch2[tid], cur);  // hash of this node's children, before the column
            if (FLOW && fs.rec) cur = rate_of(flow_perm<PACE>(fs, fidx + 3u, cur, flow_capacity4<PACE>(fs, fidx, lv), false));
            else
            cur = combine_with_column<PACE>(cur, sponge_capacity4<PACE>(lv[0], lv[1], lv[2], lv[3]));
            store_hash(xcol[tid], cur);
        }
        if (exA) buf ^= 1u;
        if (exB) __syncthreads();
        if (on && dg >= 0) {
            const uint32_t* lv = leafv + ((size_t)dg * G + j) * 8;
            uint32_t w_sib = fl[dslot * G + j] >> 16;
            if (w_sib == 0xFFFFu) {
                uint32_t e = ent[pl_ * G + j];
                if (ent_sib(e) != 0xFFu) {
                    sibh = load_hash(xcol[gbase + ent_sib(e)]);
                    if (psib) store_hash(psib + (size_t)(top - 1 - pl_) * 8, load_hash(xch2[gbase + ent_sib(e)]));
                    if (FLOW && fs.rec)  // the circuit combines sibling_hashes[i] with the sibling's column itself
                        sibh = rate_of(flow_perm<PACE>(fs, fidx + 4u, load_hash(xch2[gbase + ent_sib(e)]), flow_capacity4<PACE>(fs, fidx + 1u, lv + 4), false));
                } else bad = true;
            } else if (w_sib + 1 < L->hash_n) {
                const Hash8 wl = load_hash(w + L->hash_off + 8 * w_sib), wr = load_hash(w + L->hash_off + 8 * (w_sib + 1));
                if (hash_over(wl) | hash_over(wr)) atomicOr(&a.ctxs[p].flags, 1u << R_PARSE);
                Hash8 sn = hash_tree<PACE>(wl, wr);
                if (psib) store_hash(psib + (size_t)(top - 1 - pl_) * 8, sn);
                if (FLOW && fs.rec) sibh = rate_of(flow_perm<PACE>(fs, fidx + 4u, sn, flow_capacity4<PACE>(fs, fidx + 1u, lv + 4), false));
                else
                sibh = combine_with_column<PACE>(sn, sponge_capacity4<PACE>(lv[4], lv[5], lv[6], lv[7]));
            } else bad = true;
            have_sib = true;
            dslot++;
        }
    }
    if (live) {
        uint32_t want_hw = slot == 0 ? (uint32_t)h->wf_total : lvl_s(h->lvl[1]) - s_top;
        bool ok = !bad && want_hw == L->hash_n && (Lc || hash_eq(cur, load_hash(w + L->commit_off)));
        const uint32_t fl = (ok ? 0u : 1u << (slot == 0 ? R_FRI_FIRST : R_FRI_INNER)) | ((bad || want_hw != L->hash_n) ? F_RESCAN : 0u);
        if (fl) atomicOr(&a.ctxs[p].flags, fl);
    }
    if (Lc) {
        if constexpr (FLOW) {
            // this path's record of the step from level Lc: every column level (four extra records each) lies above the cap
            capfl[0][tid] = (live && fs.rec) ? fbase + 4u + (top - Lc) + 4u * dslot : 0xFFFFFFFFu;
            capfl[1][tid] = live ? (qj >> (M - Lc)) : 0u;
        }
        merkle_cap<VB, FLOW, PACE>(xch, capmask, capgrp, Lc, per_block, live, grp, live ? (qj >> (M - Lc)) : 0u, cur, live ? psib : nullptr, top - 2u,
                                a.Lt, a.pcapn, a.pcapm, bx * per_block, a.n, 1u + a.maxInner, slot, capfl, G);
    }
}

// ---------------------------------------------------------------- k_cap_top
// The last levels of every tree, ONE LANE PER (slot, tree).  Inside the Merkle kernels the top levels of the cap keep
// few lanes busy — 16 proofs x 2 and x 1 node positions are a half and a quarter of one wave, each at the price of a
// full wave-level permutation — and that is 8 of the 20 permutation slots per tree the cap costs a 16-query proof.  Here
// a wave carries 64 trees and 2^Lt - 1 sequential node hashes each (3 for Lt = 2, 7 for Lt = 3): the same nodes in 3
// (7) slots per tree.  Same walk as merkle_cap (same witness ranks, same flags), nodes in registers.
struct CapTopIndex {
    uint32_t first_block[MAX_FUSED + 1];  // blocks of 256 (slot, tree) items per argument set
    // Item order inside an argument set.  0: slot-major — the trees of a proof side by side, a wave touches 16 proofs.  1:
    // tree-major — a wave holds 64 proofs' copies of ONE tree: a launch under one configuration is sized for the deepest
    // trees the parser admits, and the lanes of the FRI layers no proof of the batch has then leave as whole waves; but a
    // wave's loads go to 64 proofs (64 pages) instead of 16, which a small launch — few waves, their three or seven
    // dependent hashes on the step's critical chain — pays for: 4 096 proofs 2.55 -> 2.63 ms, 8 192: 4.70 -> 4.74
    // (gpurun_out/r5_m/ab.txt, r5_p_ab.txt).  The host picks by the launch's size — a TEMPLATE argument of the cap kernels: with
    // the order chosen at run time from this struct the small launch was as slow as with tree-major (measured, same box).
};

static_assert(sizeof(Fused<MerkleArgs>) + sizeof(CapTopIndex) + 8 <= 4096 && sizeof(Fused<MerkleArgs>) + sizeof(FlowArgs) <= 4096,
              "kernel arguments are limited to 4 KB");

template <int LT, int PACE = 1>
__device__ __forceinline__ void cap_top_walk(const MerkleArgs& a, uint32_t slot_, uint32_t ti, bool pair) {
    const uint32_t T = pair ? 1u + a.maxInner : 4u;
    const size_t idx = (size_t)slot_ * T + ti;
    const bool mid = a.Lt2 != a.Lt;  // this level's nodes come from k_cap_mid
    uint32_t mask = (uint32_t)(mid ? (pair ? a.pcapm2 : a.tcapm2) : (pair ? a.pcapm : a.tcapm))[idx];
    if (!mask) return;  // a rejected proof, or a tree this proof does not have
    const uint32_t* capn = (mid ? (pair ? a.pcapn2 : a.tcapn2) : (pair ? a.pcapn : a.tcapn)) + (idx << LT) * 8;
    const uint32_t p = a.pl.proof_of(slot_);
    const ProofMeta& m = a.metas[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
    const PlanHdr& h = a.pl.hdr[slot_];
    const uint32_t *hw, *root;
    const uint16_t* wf = nullptr;
    uint32_t hw_n, s_top, fail_bit;
    if (pair) {
        const FriLayerRef& L = ti == 0 ? m.first : m.inner[ti - 1];
        const uint32_t top = ti == 0 ? m.M : m.M - ti;
        hw = w + L.hash_off; hw_n = L.hash_n; wf = ti == 0 ? h.wf : nullptr; s_top = lvl_s(h.lvl[top]);
        root = w + L.commit_off; fail_bit = ti == 0 ? R_FRI_FIRST : R_FRI_INNER;
    } else {
        const uint32_t mx = (ti == 3) ? m.M : umax(m.A, m.B);
        hw = w + m.hw_off[ti]; hw_n = m.hw_n[ti]; s_top = lvl_s(h.lvl[mx + 1]);
        root = w + W_COMMIT0 + 8 * ti; fail_bit = R_MERKLE_T0 + ti;
    }
    uint32_t* flags = &a.ctxs[p].flags;
    Hash8 node[1 << LT];
#pragma unroll
    for (int k = 0; k < (1 << LT); k++) node[k] = ((mask >> k) & 1u) ? load_hash(capn + 8 * k) : zero8();
#pragma unroll
    for (int l = LT - 1; l >= 0; l--) {  // parent level
        const uint32_t lack = (mask ^ (mask >> 1)) & 0x55555555u;  // bit 2p': exactly one child present
        const uint32_t base = wf ? (uint32_t)wf[l + 1] : lvl_s(h.lvl[l + 2]) - s_top;
        uint32_t next = 0;
#pragma unroll
        for (int pp = 0; pp < (1 << l); pp++) {
            const uint32_t pres = (mask >> (2 * pp)) & 3u;
            if (pres) {
                Hash8 left = node[2 * pp], right = node[2 * pp + 1];
                bool bad = false;
                if (pres != 3u) {
                    const uint32_t wi = base + (uint32_t)__popc(lack & ((1u << (2 * pp)) - 1u));
                    Hash8 w8 = zero8();
                    if (wi < hw_n) {
                        w8 = load_hash(hw + 8 * wi);
                        if (hash_over(w8)) atomicOr(flags, 1u << R_PARSE);
                    } else bad = true;
                    if (!(pres & 1u)) left = w8;
                    if (!(pres & 2u)) right = w8;
                }
                node[pp] = hash_tree<PACE>(left, right);
                if (l == 0) {
                    if (bad || !hash_eq(node[0], load_hash(root))) atomicOr(flags, (1u << fail_bit) | (bad ? F_RESCAN : 0u));
                } else if (bad) atomicOr(flags, (1u << fail_bit) | F_RESCAN);
                next |= 1u << pp;
            }
        }
        mask = next;
    }
}

template <int PACE = 1, bool TREE_MAJOR = false>
__global__ __launch_bounds__(256) void k_cap_top(Fused<MerkleArgs> f, CapTopIndex ix, uint32_t pair) {
    RSV_TAG(pair ? 4 : 3);
    uint32_t k = 0;
    while (k + 1 < f.nb && blockIdx.x >= ix.first_block[k + 1]) k++;
    const MerkleArgs& a = f.a[k];
    const uint32_t T = pair ? 1u + a.maxInner : 4u;
    const uint64_t item = (uint64_t)(blockIdx.x - ix.first_block[k]) * 256 + threadIdx.x;
    if (!a.Lt || item >= (uint64_t)a.n * T) return;
    const uint32_t ti = TREE_MAJOR ? (uint32_t)(item / a.n) : (uint32_t)(item % T);
    const uint32_t slot_ = TREE_MAJOR ? (uint32_t)(item % a.n) : (uint32_t)(item / T);
    if (a.Lt2 == 2) cap_top_walk<2, PACE>(a, slot_, ti, pair != 0);
    else cap_top_walk<3, PACE>(a, slot_, ti, pair != 0);
}

// ---------------------------------------------------------------- k_cap_mid
// The cap levels between the tree kernels' hand-over (level Lt = Lc) and k_cap_top's (level Lt2), ONE LANE PER
// (slot, tree, level-Lt2 position): the lane owns the subtree below its position — 2^LA nodes of level Lt in registers,
// 2^LA - 1 sequential node hashes — and leaves the subtree's root in capn2.  For buckets whose in-kernel cap levels fill
// their waves badly (three 80-query proofs per workgroup: 96, 48, 24 lanes in 2 + 1 + 1 waves): here the lanes are dealt
// over the whole launch, every wave full.  Witness ranks count the lacking parents of the WHOLE level (as merkle_cap: the
// batched order of components/hints/src/decommit.rs:91-139), so every lane carries the level's presence mask along.
__device__ __forceinline__ unsigned long long cap_parent_mask(unsigned long long cm) {  // bit p = bit 2p | bit 2p + 1
    unsigned long long t = (cm | (cm >> 1)) & 0x5555555555555555ull;
    t = (t | (t >> 1)) & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0f0f0f0f0f0f0f0full;
    t = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;
    t = (t | (t >> 8)) & 0x0000ffff0000ffffull;
    t = (t | (t >> 16)) & 0x00000000ffffffffull;
    return t;
}

template <int LA, int PACE = 1>
__device__ __forceinline__ void cap_mid_walk(const MerkleArgs& a, uint32_t slot_, uint32_t ti, uint32_t sub, bool pair) {
    const uint32_t T = pair ? 1u + a.maxInner : 4u;
    const size_t idx = (size_t)slot_ * T + ti;
    unsigned long long cm = (pair ? a.pcapm : a.tcapm)[idx];  // presence at level Lt
    unsigned long long* capm2 = (pair ? a.pcapm2 : a.tcapm2) + idx;
    if (!cm) {  // a rejected proof, or a tree this proof does not have
        if (sub == 0) *capm2 = 0ull;
        return;
    }
    const uint32_t Lt = a.Lt, Lt2 = a.Lt2;
    const uint32_t* capn = (pair ? a.pcapn : a.tcapn) + (idx << Lt) * 8;
    const uint32_t p = a.pl.proof_of(slot_);
    const ProofMeta& m = a.metas[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
    const PlanHdr& h = a.pl.hdr[slot_];
    const uint32_t* hw;
    const uint16_t* wf = nullptr;
    uint32_t hw_n, s_top, fail_bit;
    if (pair) {
        const FriLayerRef& L = ti == 0 ? m.first : m.inner[ti - 1];
        const uint32_t top = ti == 0 ? m.M : m.M - ti;
        hw = w + L.hash_off; hw_n = L.hash_n; wf = ti == 0 ? h.wf : nullptr; s_top = lvl_s(h.lvl[top]);
        fail_bit = ti == 0 ? R_FRI_FIRST : R_FRI_INNER;
    } else {
        const uint32_t mx = (ti == 3) ? m.M : umax(m.A, m.B);
        hw = w + m.hw_off[ti]; hw_n = m.hw_n[ti]; s_top = lvl_s(h.lvl[mx + 1]);
        fail_bit = R_MERKLE_T0 + ti;
    }
    uint32_t* flags = &a.ctxs[p].flags;
    const uint32_t first = sub << LA;  // this subtree's first position at level Lt
    const bool any = ((cm >> first) & ((1ull << (1 << LA)) - 1ull)) != 0ull;
    Hash8 node[1 << LA];
#pragma unroll
    for (int k = 0; k < (1 << LA); k++) node[k] = ((cm >> (first + k)) & 1ull) ? load_hash(capn + 8 * (first + k)) : zero8();
#pragma unroll
    for (int s = LA - 1; s >= 0; s--) {  // the subtree has 2^s parents at level l = Lt2 + s; cm = presence at child level l + 1
        const uint32_t l = Lt2 + (uint32_t)s;
        const unsigned long long lack = (cm ^ (cm >> 1)) & 0x5555555555555555ull;  // bit 2p': exactly one child present
        const uint32_t base = wf ? (uint32_t)wf[l + 1] : lvl_s(h.lvl[l + 2]) - s_top;
        if (any) {
#pragma unroll
            for (int pp = 0; pp < (1 << s); pp++) {
                const uint32_t gp = (sub << s) + (uint32_t)pp;  // parent position in its level
                const uint32_t pres = (uint32_t)(cm >> (2 * gp)) & 3u;
                if (pres) {
                    Hash8 left = node[2 * pp], right = node[2 * pp + 1];
                    if (pres != 3u) {
                        const uint32_t wi = base + (uint32_t)__popcll(lack & ((1ull << (2 * gp)) - 1ull));
                        Hash8 w8 = zero8();
                        if (wi < hw_n) {
                            w8 = load_hash(hw + 8 * wi);
                            if (hash_over(w8)) atomicOr(flags, 1u << R_PARSE);
                        } else atomicOr(flags, (1u << fail_bit) | F_RESCAN);
                        if (!(pres & 1u)) left = w8;
                        if (!(pres & 2u)) right = w8;
                    }
                    node[pp] = hash_tree<PACE>(left, right);
                }
            }
        }
        cm = cap_parent_mask(cm);
    }
    if (sub == 0) *capm2 = cm;
    if (any) store_hash((pair ? a.pcapn2 : a.tcapn2) + ((idx << Lt2) + sub) * 8, node[0]);
}

template <int PACE = 1, bool TREE_MAJOR = false>
__global__ __launch_bounds__(256) void k_cap_mid(Fused<MerkleArgs> f, CapTopIndex ix, uint32_t pair) {
    RSV_TAG(pair ? 4 : 3);
    uint32_t k = 0;
    while (k + 1 < f.nb && blockIdx.x >= ix.first_block[k + 1]) k++;
    const MerkleArgs& a = f.a[k];
    if (a.Lt2 == a.Lt) return;  // (such an argument set has no blocks here)
    const uint32_t T = pair ? 1u + a.maxInner : 4u;
    const uint64_t item = (uint64_t)(blockIdx.x - ix.first_block[k]) * 256 + threadIdx.x;
    if (item >= ((uint64_t)a.n * T << a.Lt2)) return;
    const uint32_t sub = (uint32_t)(item & ((1u << a.Lt2) - 1u));
    const uint64_t tree = item >> a.Lt2;
    const uint32_t ti = TREE_MAJOR ? (uint32_t)(tree / a.n) : (uint32_t)(tree % T);  // (item order: CapTopIndex)
    const uint32_t slot_ = TREE_MAJOR ? (uint32_t)(tree % a.n) : (uint32_t)(tree / T);
    if (a.Lt - a.Lt2 == 2) cap_mid_walk<2, PACE>(a, slot_, ti, sub, pair != 0);
    else cap_mid_walk<3, PACE>(a, slot_, ti, sub, pair != 0);
}

}  // namespace rsv
