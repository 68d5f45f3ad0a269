// k_query.hpp — DEEP quotients, FRI folds, last-layer evaluation per query (k_query) and the arithmetic probes.  Part of the pipeline described in verify.hpp.
#pragma once
#include "k_plan.hpp"

namespace rsv {

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
    // optional value dump (rsv_hints_out::d_query_values), [n][G][qv_stride] in transcript query order, per query:
    //   answers[3] | first-layer folds[3] | value entering inner layer i, i < maxInner | value entering the last-layer
    //   check | the last-layer polynomial evaluated at the query's point      (QM31 each; absent groups stay zero)
    uint32_t* qv_out;
    uint32_t qv_stride;
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
// No array here is indexed at run time: the factors are brought into index-bit order by a 4-stage barrel shift of
// registers.  (A run-time index put d[] in scratch memory, and a kernel that uses scratch costs ~15 us more per launch on
// this part — tools/chain_lab.hip — which a small batch pays in full: k_query is on its chain of dependent kernels.)
// hi0, hi_step: this call's share of the blocks of 16 coefficients (the row form of k_query splits them over 16 threads and
// adds the partial sums up; default: all of them)
__device__ __forceinline__ QM31 line_eval(const uint32_t* __restrict__ cf, uint32_t log_n, uint32_t n, uint32_t x, uint32_t hi0 = 0,
                                          uint32_t hi_step = 1) {
    // r[15 - j] = pi^j(x); index bit b of a coefficient pairs with pi^(log_n-1-b)(x) = r[b + 16 - log_n]
    uint32_t r[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { r[15 - j] = x; x = m_sub(m_dbl(m_sqr(x)), 1u); }
    const uint32_t s = (16u - log_n) & 15u;  // log_n = 0: no factor is read
#pragma unroll
    for (int st = 8; st >= 1; st >>= 1) {
        const bool on = (s & (uint32_t)st) != 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint32_t from = t + st < 16 ? r[t + st] : 1u;
            r[t] = on ? from : r[t];
        }
    }
    // now r[b] is the factor of index bit b (b < log_n)
    const uint32_t nlo = log_n < 4 ? log_n : 4u;
    uint32_t wl[16];
#pragma unroll
    for (int t = 0; t < 16; t++) wl[t] = 1u;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t db = (uint32_t)b < nlo ? r[b] : 1u;
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (t & (1 << b)) wl[t] = m_mul(wl[t], db);
    }
    QM31 acc = q_zero();
    const uint32_t n_hi = n >> nlo, n_lo = 1u << nlo;
#pragma unroll 1
    for (uint32_t hi = hi0; hi < n_hi; hi += hi_step) {
        // hi > 0 only when log_n > 4, i.e. nlo = 4: bit b of hi is index bit b + 4
        uint32_t wh = 1u;
#pragma unroll
        for (int b = 0; b < 12; b++)
            if ((uint32_t)b + 4u < log_n && ((hi >> b) & 1u)) wh = m_mul(wh, r[b + 4]);
        QM31 inner = q_zero();
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if ((uint32_t)t < n_lo) inner = q_add(inner, q_mul_m(ldq(cf + 4 * ((hi << nlo) + t)), wl[t]));
            // four coefficients in flight, not sixteen: the scheduler would otherwise hoist all 64 words of loads and
            // the caller (k_query) would lose a wave per SIMD to this function's registers
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        acc = q_add(acc, q_mul_m(inner, wh));
    }
    return acc;
}

// folding/src/lib.rs:194-204: the folded value of a query must equal the last-layer polynomial at its point
__device__ __forceinline__ bool last_layer_ok(QM31 eval, QM31 folded) { return q_eq(eval, folded); }

// QB: columns per batch of loads (1, 2 or 4: a divisor of the interaction columns' period).  4 is the latency form (small
// batches, where this kernel is on the step's chain of dependent kernels: 1 024 proofs 1.556 -> 1.532 ms) at 104-114
// registers = 4 waves per SIMD; 2 keeps 96 registers = 5 waves per SIMD for the large batches that run it underneath the
// trace trees (65 536 proofs: 34.13 ms against 34.87 with 4).
// ROW: the kernel on VIRTUAL lanes, as the row form of the tree kernels (k_merkle.hpp) — a DPP row of 16 threads per query.
// They compute the same indices and the same folds; what is a sum they split: the 142 columns of the quotients' linear
// combinations and the blocks of the last-layer polynomial go sixteen ways and are added up over the row.  For launches
// of a few waves (with the trees' row form): one proof's k_query 95 -> 47 us (of the 95: last layer 42, quotients 33, inner layers 20), 128 proofs 112 -> 71 us.
template <int BLOCK, uint32_t QB, bool ROW = false>
__global__ __launch_bounds__(BLOCK) void k_query(Fused<QueryArgs> f) {
    constexpr int VB = ROW ? BLOCK / 16 : BLOCK;                       // lanes of this kernel's indexing per workgroup
    const uint32_t tid = ROW ? threadIdx.x >> 4 : threadIdx.x;         // the lane of the indexing below
    const uint32_t part = ROW ? threadIdx.x & 15u : 0u;                // ROW: this thread's sixteenth of a query's sums
    // sum of a QM31 over the 16 threads of a row, in every one of them
    auto row_sum_q = [](QM31 v) { return q_mk(sum_row(v.a.a), sum_row(v.a.b), sum_row(v.b.a), sum_row(v.b.b)); };
    __shared__ uint32_t xq[VB][4];
    // each lane's first-layer folds (one QM31 per column size), word-major so that a wave's access is conflict free.  In
    // LDS, not registers: they are written early and read in the inner-layer loop, and 12 more live registers would cost
    // the kernel its fifth wave per SIMD.  (Indexed arrays in registers end up in scratch memory, which costs a small
    // batch ~15 us per launch: tools/chain_lab.hip.)
    __shared__ uint32_t fst[3][4][VB];
    __shared__ uint32_t dps[3][2][VB];  // and its domain point at each column size (x, y), for the same reason
    // The FRI trees wait for this kernel while it shares the machine with the trace trees: its waves go first in the
    // SIMDs' arbitration, so that it is over in a fraction of the trace trees' time and the FRI trees start beside them.
    __builtin_amdgcn_s_setprio(3);
    RSV_FUSED_SELECT(f, a, bx);
    const uint32_t G = a.pl.G, per_block = VB / G;
    const uint32_t grp = tid / G, j = tid % G;
    const uint32_t slot = bx * per_block + grp;
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
    uint32_t* qv = (live && a.qv_out) ? a.qv_out + ((size_t)slot * G + c->qperm[j]) * a.qv_stride : nullptr;
    uint32_t n_sizes = live ? c->n_sizes : 0;
    auto put_first = [&](uint32_t g, QM31 v) {
        fst[g][0][tid] = v.a.a; fst[g][1][tid] = v.a.b; fst[g][2][tid] = v.b.a; fst[g][3][tid] = v.b.b;
    };
    auto get_first = [&](uint32_t g) { return q_mk(fst[g][0][tid], fst[g][1][tid], fst[g][2][tid], fst[g][3][tid]); };
    // Domain points.  One scalar multiplication gives the point of the query at level M; the points at the
    // smaller column sizes follow by the doubling map pi(x, y) = (2x^2 - 1, 2xy): doubling the level-l point of
    // position pos gives the level-(l-1) point of pos >> 1 up to the sign of y, which is fixed by bit 0 of the
    // respective positions (CanonicCoset::circle_domain().at(bit_reverse(.))).
    if (live) {
        CPoint cur = domain_point(M, qj);
        uint32_t lvl = M;
        auto descend = [&](uint32_t g) {
            const uint32_t l = g < n_sizes ? c->sizes[g] : lvl;
            while (lvl > l) {
                const uint32_t pos = qj >> (M - lvl);
                uint32_t y2 = m_dbl(m_mul(cur.x, cur.y));
                cur.x = m_sub(m_dbl(m_sqr(cur.x)), 1u);
                cur.y = ((pos ^ (pos >> 1)) & 1u) ? m_neg(y2) : y2;
                lvl--;
            }
            dps[g][0][tid] = cur.x; dps[g][1][tid] = cur.y;
        };
        descend(0); descend(1); descend(2);
    }
    // ---- DEEP quotients + first-layer fold, per column log size
    for (uint32_t g = 0; g < 3; g++) {
        QM31 answer = q_zero();
        uint32_t l = 0, pos = 0;
        bool on = live && g < n_sizes;
        uint32_t dpg_y = 0;
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
                    // Four columns at a time, every load of the four issued before the first product: one load per
                    // iteration is consumed at once, and the lane then pays the whole memory latency 134 times (a single
                    // proof: 0.11 ms of a 1.3 ms call).  Interaction columns 4-7 / 12-15 (k & 4) carry a second sample
                    // point: a batch starts at a multiple of 4, so a whole batch has one or none.
                    if constexpr (ROW) {
                        // this thread's columns: part, part + 16, ...  Column k of the segment reads alpha power col + k; in an
                        // interaction tree the columns with (k & 4) carry the second sample point, whose powers are numbered
                        // in column order: 4 * (k >> 3) + (k & 3) such columns come before column k
                        for (uint32_t k = part; k < nc; k += 16u) {
                            const uint32_t v = inb ? qv[off + k] : 0u;
                            r0 = q_add(r0, q_mul_m(ldq(c->apow[col + k]), v));
                            if (t == 2 && (k & 4u)) r1 = q_add(r1, q_mul_m(ldq(c->apow[ncols_group + dbl + 4u * (k >> 3) + (k & 3u)]), v));
                        }
                        col += nc;
                        if (t == 2) dbl += 4u * (nc >> 3) + ((nc & 7u) > 4u ? (nc & 7u) - 4u : 0u);
                    } else
                    for (uint32_t k0 = 0; k0 < nc; k0 += QB) {
                        const bool twice = t == 2 && (k0 & 4);
                        uint32_t vv[QB];
                        QM31 aa[QB];
#pragma unroll
                        for (uint32_t u = 0; u < QB; u++) {
                            const bool ok = k0 + u < nc;
                            vv[u] = (ok && inb) ? qv[off + k0 + u] : 0u;          // (a column beyond nc contributes 0)
                            aa[u] = ldq(c->apow[ok ? col + u : col]);
                        }
#pragma unroll
                        for (uint32_t u = 0; u < QB; u++) r0 = q_add(r0, q_mul_m(aa[u], vv[u]));
                        if (twice) {  // the second sample point's powers in the same registers (keeps the kernel at 5 waves per SIMD)
#pragma unroll
                            for (uint32_t u = 0; u < QB; u++) aa[u] = ldq(c->apow[ncols_group + dbl + (k0 + u < nc ? u : 0u)]);
#pragma unroll
                            for (uint32_t u = 0; u < QB; u++) r1 = q_add(r1, q_mul_m(aa[u], vv[u]));
                        }
                        const uint32_t done = nc - k0 < QB ? nc - k0 : QB;
                        col += done;
                        if (twice) dbl += done;
                    }
                }
            }
            if constexpr (ROW) { r0 = row_sum_q(r0); r1 = row_sum_q(r1); }  // the sixteen threads' shares
            const CPoint dpg = {dps[g][0][tid], dps[g][1][tid]};  // read here, not above: not live over the columns
            dpg_y = dpg.y;
            for (uint32_t bi = 0; bi < c->n_batches[g]; bi++) {
                const QBatch& qb = c->batch[g][bi];
                CM31 prx = c_mk(qb.prx[0], qb.prx[1]), pix = c_mk(qb.pix[0], qb.pix[1]);
                CM31 pry = c_mk(qb.pry[0], qb.pry[1]), piy = c_mk(qb.piy[0], qb.piy[1]);
                CM31 den = c_sub(c_mul(c_sub(prx, c_mk(dpg.x, 0)), piy), c_mul(c_sub(pry, c_mk(dpg.y, 0)), pix));
                const CM31 inv_den = c_inv(den);
                __builtin_amdgcn_sched_barrier(0);  // the inversion's temporaries are dead before the numerator's loads are issued (96 registers)
                QM31 num = q_sub(q_mul_c(bi ? r1 : r0, piy), q_add(q_mul_m(ldq(qb.sa), dpg.y), ldq(qb.sb)));
                answer = q_add(answer, q_mul_c(num, inv_den));
            }
        }
        // exchange answers: the pair sibling may be another query of this proof
        stq(xq[tid], answer);
        __syncthreads();
        if (on) {
            uint32_t e = ent[l * G + j];
            QM31 sib;
            if (ent_sib(e) != 0xFFu) sib = ldq(xq[gbase + ent_sib(e)]);
            else {
                uint32_t wi = c->fw_base[g] + ent_lb(e);
                sib = wi < m->first.wit_n ? ldq(w + m->first.wit_off + 4 * wi) : q_zero();
                if ((sib.a.a >= P) | (sib.a.b >= P) | (sib.b.a >= P) | (sib.b.b >= P)) flags |= 1u << R_PARSE;  // fri_witness is read here (layout.hpp)
            }
            uint32_t* lv = leafv + ((size_t)g * G + j) * 8;
            stq(lv, answer); stq(lv + 4, sib);
            // fold circle -> line with 1/y of the pair's base point (folding/src/lib.rs:57-90); the base
            // point (bit 0 of the position cleared) is the conjugate of this point when the position is odd
            uint32_t by = (pos & 1u) ? m_neg(dpg_y) : dpg_y;
            const QM31 fg = fold_pair(answer, sib, pos & 1u, m_inv(by), ldq(c->fri_alpha[M - l]));
            put_first(g, fg);
            if (a.folded_out) stq(a.folded_out + (((size_t)slot * 3 + g) * G + c->qperm[j]) * 4, fg);
            if (qv) { stq(qv + 4 * g, answer); stq(qv + 12 + 4 * g, fg); }
        }
        __syncthreads();
    }
    // ---- inner layers (folding/src/lib.rs:120-192)
    // x-coordinate of the pair base point at line-domain level l: X_{M-1} = +-x_M and X_{l-1} = +-(2 X_l^2 - 1)
    // (half_odds(l).at(i) doubles to half_odds(l-1).at(i mod 2^(l-1)); clearing bit 0 of an odd position moves
    // the bit-reversed index by half the coset = the point (-1, 0), i.e. negates x).
    QM31 folded = q_zero();
    uint32_t l = M;
    uint32_t X = live ? dps[0][0][tid] : 0u;  // x of the query's point at the largest column size
    for (uint32_t i = 0; i < a.maxInner; i++) {
        bool on = live && i < m->n_inner;
        if (on) {
            for (uint32_t g = 0; g < n_sizes; g++)
                if (c->sizes[g] == l) {
                    QM31 al = ldq(c->fri_alpha[i]);
                    folded = q_add(q_mul(q_mul(al, al), folded), get_first(g));
                }
            l -= 1;
        }
        stq(xq[tid], folded);
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
                if ((sib.a.a >= P) | (sib.a.b >= P) | (sib.b.a >= P) | (sib.b.b >= P)) flags |= 1u << R_PARSE;
            }
            // a witness list of the wrong length is not read completely: k_rescan reads the whole proof (layout.hpp)
            if (j == 0 && lvl_tl(h->lvl[l]) != L.wit_n) flags |= (1u << R_FRI_INNER) | F_RESCAN;  // hints/src/folding.rs:558
            uint32_t* lv = leafv + ((size_t)(3 + i) * G + j) * 8;
            stq(lv, folded); stq(lv + 4, sib);
            if (qv) stq(qv + 24 + 4 * i, folded);
            uint32_t xr = (i == 0) ? X : m_sub(m_dbl(m_sqr(X)), 1u);
            X = (pos & 1u) ? m_neg(xr) : xr;
            folded = fold_pair(folded, sib, pos & 1u, m_inv(X), ldq(c->fri_alpha[i + 1]));
        }
        __syncthreads();
    }
    // ---- last layer (folding/src/lib.rs:194-204, primitives/line/src/lib.rs:39-67)
    if (live) {
        // x of half_odds(l-1).at(bit_reverse(pos >> 1)) = pi(X_l) (for a proof without inner layers: pi of x_M)
        QM31 acc = line_eval(w + m->last_off, m->log_last, m->last_n, m_sub(m_dbl(m_sqr(X)), 1u), part, ROW ? 16u : 1u);
        if constexpr (ROW) acc = row_sum_q(acc);
        if (!last_layer_ok(acc, folded)) flags |= 1u << R_FRI_LAST;
        if (qv) { stq(qv + 24 + 4 * a.maxInner, folded); stq(qv + 28 + 4 * a.maxInner, acc); }
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

// rsv_last_layer_check: the comparison k_query raises RSV_R_FRI_LAST from, on its own: ok[i] = 1 iff the polynomial
// evaluated at x[i] equals folded[i].
__global__ __launch_bounds__(256) void k_last_layer_check(const uint32_t* __restrict__ coeffs, uint32_t log_n,
                                                           const uint32_t* __restrict__ x, const uint32_t* __restrict__ folded,
                                                           uint8_t* __restrict__ ok, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ok[i] = last_layer_ok(line_eval(coeffs, log_n, 1u << log_n, x[i]), ldq(folded + 4 * i)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_line_eval(const uint32_t* __restrict__ coeffs, uint32_t log_n,
                                                    const uint32_t* __restrict__ x, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stq(out + 4 * i, line_eval(coeffs, log_n, 1u << log_n, x[i]));
}

}  // namespace rsv
