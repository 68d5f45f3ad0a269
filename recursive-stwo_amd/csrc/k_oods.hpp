// k_oods.hpp — logup sum and OODS composition identity (k_oods).  Part of the pipeline described in verify.hpp.
#pragma once
#include "verify_common.hpp"

namespace rsv {

// ------------------------------------------------------------------- k_oods
// Logup total-sum check (fiat_shamir/src/lib.rs:133-141) and the OODS
// composition identity (components/recursive/composition/src/**).
struct EvalCtx {
    QM31 rc, acc, dinv, z, alpha, alpha2, shift;
    QM31 fp[5], fq[5];
    const uint32_t* w;
    int inter;  // next interaction sample index
    uint32_t prefix_diff = 0;  // OR of (word ^ expected) over the sample-count prefixes passed so far (bincode shape: k_parse.hpp)
    // a sampled value, and — with the first sample of a column — the column's u64 sample-count prefix in front of it (the same
    // cache line): the evaluation reads all 142 samples, so it passes all 134 column prefixes
    __device__ QM31 smp(int k) {
        const uint32_t po = SAMPLES.pre_of[k];
        if (po) prefix_diff |= (w[po] ^ (uint32_t)SAMPLES.cnt_of[k]) | w[po + 1];
        return ldq(w + SAMPLES.off[k]);
    }
    // data_structures.rs:26-28,166-169
    __device__ void constraint(QM31 v) { acc = q_add(q_mul(acc, rc), q_mul(v, dinv)); }
    // data_structures.rs:147-164.  K (the fraction's index) and the batch geometry below are compile-time constants so
    // that fp / fq stay in registers (a run-time index would send them to scratch memory).
    template <int K> __device__ __forceinline__ void relation(QM31 mult, QM31 v0, QM31 v1) {
        fp[K] = mult;
        fq[K] = q_sub(q_add(v0, q_mul(alpha, v1)), z);
    }
    template <int K> __device__ __forceinline__ void relation(QM31 mult, QM31 v0, QM31 v1, QM31 v2) {
        fp[K] = mult;
        fq[K] = q_sub(q_add(q_add(v0, q_mul(alpha, v1)), q_mul(alpha2, v2)), z);
    }
    // data_structures.rs:171-210: NF fractions in batches of BATCH
    template <int NF, int BATCH> __device__ __forceinline__ void finalize_logup() {
        constexpr int NB = (NF + BATCH - 1) / BATCH;
        QM31 prev = q_zero();
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const int lo = bi * BATCH, hi = lo + BATCH < NF ? lo + BATCH : NF;
            QM31 pp = fp[lo], qq = fq[lo];
#pragma unroll
            for (int k = lo + 1; k < hi; k++) {
                pp = q_add(q_mul(pp, fq[k]), q_mul(fp[k], qq));
                qq = q_mul(qq, fq[k]);
            }
            if (bi < NB - 1) {
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
// The 16 QM31 words of the Poseidon AIR's round state, one lane's copy in LDS: word-major over the workgroup's 64 lanes
// (st[(4 i + c) * 64 + lane]: conflict-free), so that the matrix steps can be out-of-line functions walking it with
// run-time indices — as a register array behind a pointer it lived in scratch memory (1 440 B per lane, 1.5 GB of HBM
// traffic per 65 536 proofs for a stage that reads 0.15 GB of samples).
struct QState {
    uint32_t* base;  // this lane's word 0
    __device__ __forceinline__ QM31 get(int i) const {
        return q_mk(base[(4 * i) * 64], base[(4 * i + 1) * 64], base[(4 * i + 2) * 64], base[(4 * i + 3) * 64]);
    }
    __device__ __forceinline__ void set(int i, QM31 v) const {
        base[(4 * i) * 64] = v.a.a; base[(4 * i + 1) * 64] = v.a.b; base[(4 * i + 2) * 64] = v.b.a; base[(4 * i + 3) * 64] = v.b.b;
    }
};
constexpr int QSTATE_WORDS = 3 * 16 * 4 * 64;  // per workgroup of 64 lanes: the round state, and the 16 `in` and 16 `out` samples
// poseidon.rs:12-71 over QM31
__device__ inline void q_m4(QState s, int g) {
    const QM31 x0 = s.get(g), x1 = s.get(g + 1), x2 = s.get(g + 2), x3 = s.get(g + 3);
    QM31 t0 = q_add(x0, x1), t02 = q_dbl(t0), t1 = q_add(x2, x3), t12 = q_dbl(t1);
    QM31 t2 = q_add(q_dbl(x1), t1), t3 = q_add(q_dbl(x3), t0);
    QM31 t4 = q_add(q_dbl(t12), t3), t5 = q_add(q_dbl(t02), t2);
    s.set(g, q_add(t3, t5)); s.set(g + 1, t5); s.set(g + 2, q_add(t2, t4)); s.set(g + 3, t4);
}
__device__ __noinline__ void q_external(QState s) {
#pragma unroll 1
    for (int g = 0; g < 4; g++) q_m4(s, 4 * g);
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        const QM31 a = s.get(j), b = s.get(j + 4), c = s.get(j + 8), d = s.get(j + 12);
        const QM31 sum = q_add(q_add(a, b), q_add(c, d));
        s.set(j, q_add(a, sum)); s.set(j + 4, q_add(b, sum)); s.set(j + 8, q_add(c, sum)); s.set(j + 12, q_add(d, sum));
    }
}
__device__ __noinline__ void q_internal(QState s) {
    const QM31 s0 = s.get(0);
    QM31 sum = s0;
#pragma unroll 1
    for (int i = 1; i < 16; i++) sum = q_add(sum, s.get(i));
    s.set(0, q_add(s0, q_add(q_dbl(s0), sum)));
#pragma unroll 1
    for (int i = 1; i < 16; i++) s.set(i, q_add(q_mul_m(s.get(i), 1u << (i + 1)), sum));
}

// The 86-constraint accumulator and the value it must equal (composition/src/lib.rs:60-120) for one proof whose
// sampled values sit at their fixed word offsets behind `w`.  Shared by k_oods and by the probe k_oods_probe
// (rsv_oods_eval), so that the evaluation can be checked on its own, on arbitrary samples.
// st: this lane's QState (LDS of the calling kernel: QSTATE_WORDS words per 64 lanes).  Behind the round state the same LDS
// keeps the 16 input and 16 output words of the Poseidon row, which the constraints visit four times each: 65 536 lanes
// hold 223 MB of sampled values between them, so a line left for a moment is gone from the L2 when its lane comes back
// (round 5: 0.78 GB of traffic with one global read per visit).
__device__ __forceinline__ void oods_eval(const uint32_t* w, uint32_t lp, uint32_t lq, QM31 plonk_sum, QM31 poseidon_sum, QM31 z,
                                 QM31 alpha, QM31 rc, QM31 ox, QState st, QM31& acc_out, QM31& expected_out, uint32_t& prefix_diff_out) {
    EvalCtx e;
    e.rc = rc; e.acc = q_zero(); e.z = z; e.alpha = alpha; e.alpha2 = q_mul(alpha, alpha); e.w = w;
    const QM31 one = q_one();
    {  // plonk.rs:8-82 — preprocessed samples 0..10, trace samples 50..62, interaction samples 110..122
        e.dinv = q_inv(q_double_x(ox, lp - 1));  // coset_vanishing: composition/src/lib.rs:18-29
        e.shift = q_mul_m(plonk_sum, m_inv(1u << lp));  // data_structures.rs:67-68
        e.inter = S_T2;
        const int pre = S_T0, tr = S_T1;
        QM31 enforce = e.smp(pre + 9), op = e.smp(pre + 3);
        e.constraint(q_mul(enforce, e.smp(tr + 9)));
        e.constraint(q_mul(enforce, e.smp(tr + 10)));
        e.constraint(q_mul(enforce, e.smp(tr + 11)));
        QM31 a = q_combine_ef(e.smp(tr + 0), e.smp(tr + 1), e.smp(tr + 2), e.smp(tr + 3));
        QM31 b = q_combine_ef(e.smp(tr + 4), e.smp(tr + 5), e.smp(tr + 6), e.smp(tr + 7));
        QM31 cc = q_combine_ef(e.smp(tr + 8), e.smp(tr + 9), e.smp(tr + 10), e.smp(tr + 11));
        e.constraint(q_sub(q_sub(cc, q_mul(op, q_add(a, b))), q_mul(q_mul(q_sub(one, op), a), b)));
        e.relation<0>(e.smp(pre + 4), a, e.smp(pre + 0));
        e.relation<1>(e.smp(pre + 5), b, e.smp(pre + 1));
        e.relation<2>(e.smp(pre + 6), cc, e.smp(pre + 2));
        e.relation<3>(q_neg(e.smp(pre + 8)), e.smp(pre + 7), a, b);
        e.finalize_logup<4, 2>();
    }
    {  // poseidon.rs:73-241 — preprocessed 10..50, trace 62..110, interaction samples 122..134
        e.dinv = q_inv(q_double_x(ox, lq - 1));
        e.shift = q_mul_m(poseidon_sum, m_inv(1u << lq));
        e.inter = S_T2 + 12;
        const int pre = S_T0 + 10, in = S_T1 + 12, mid = in + 16, out = in + 32;
        const int rc0 = pre + 4, rc1 = pre + 20;
        QM31 is_first = e.smp(pre), is_last = e.smp(pre + 1), is_full = e.smp(pre + 2), round_id = e.smp(pre + 3);
        QM31 not_first = q_sub(one, is_first), not_last = q_sub(one, is_last), is_partial = q_sub(not_first, is_full);
        QM31 swap_val = e.smp(mid), one_minus_swap = q_sub(one, swap_val);
        const QState inq{st.base + 16 * 4 * 64}, outq{st.base + 2 * 16 * 4 * 64};
#pragma unroll 1
        for (int i = 0; i < 16; i++) { inq.set(i, e.smp(in + i)); outq.set(i, e.smp(out + i)); }  // the one global read of each
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            QM31 lo = inq.get(i & 7), hi = inq.get((i & 7) + 8);
            st.set(i, i < 8 ? q_add(q_mul(lo, one_minus_swap), q_mul(hi, swap_val))
                            : q_add(q_mul(lo, swap_val), q_mul(hi, one_minus_swap)));
        }
        q_external(st);
#pragma unroll 1
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_first, q_sub(st.get(i), outq.get(i))));
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            QM31 full = q_pow5(q_add(inq.get(i), e.smp(rc0 + i)));
            QM31 mi = e.smp(mid + i);
            e.constraint(q_mul(is_full, q_sub(mi, full)));
            st.set(i, mi);
        }
        q_external(st);
#pragma unroll 1
        for (int i = 0; i < 16; i++) st.set(i, q_pow5(q_add(st.get(i), e.smp(rc1 + i))));
        q_external(st);
#pragma unroll 1
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_full, q_sub(outq.get(i), st.get(i))));
#pragma unroll 1
        for (int i = 0; i < 16; i++) st.set(i, inq.get(i));
#pragma unroll 1
        for (int r = 0; r < 14; r++) {
            QM31 v = q_pow5(q_add(st.get(0), e.smp(rc0 + r)));
            QM31 mi = e.smp(mid + r);
            e.constraint(q_mul(is_partial, q_sub(mi, v)));
            st.set(0, mi);
            q_internal(st);
        }
#pragma unroll 1
        for (int i = 0; i < 16; i++) e.constraint(q_mul(is_partial, q_sub(outq.get(i), st.get(i))));
        QM31 ext1 = e.smp(pre + 36), ext2 = e.smp(pre + 37), ext1_nz = e.smp(pre + 38), ext2_nz = e.smp(pre + 39);
        QM31 in_left = q_dbl(round_id), in_right = q_add(in_left, one), out_left = q_add(in_right, one),
             out_right = q_add(out_left, one);
#define EF4(q, k) q_combine_ef(q.get(k), q.get((k) + 1), q.get((k) + 2), q.get((k) + 3))
        e.relation<0>(q_sub(q_mul(ext1_nz, is_first), not_first), q_add(q_mul(is_first, ext1), q_mul(not_first, in_left)),
                   EF4(inq, 0), EF4(inq, 4));
        e.relation<1>(q_sub(q_mul(ext2_nz, is_first), not_first), q_add(q_mul(is_first, ext2), q_mul(not_first, in_right)),
                   EF4(inq, 8), EF4(inq, 12));
        e.relation<2>(q_add(q_mul(ext1_nz, is_last), not_last), q_add(q_mul(is_last, ext1), q_mul(not_last, out_left)),
                   EF4(outq, 0), EF4(outq, 4));
        e.relation<3>(q_add(q_mul(ext2_nz, is_last), not_last), q_add(q_mul(is_last, ext2), q_mul(not_last, out_right)),
                   EF4(outq, 8), EF4(outq, 12));
#undef EF4
        e.relation<4>(q_mul(is_first, not_last), swap_val, e.smp(rc0));
        e.finalize_logup<5, 3>();
    }
    {  // composition/src/lib.rs:106-120
        QM31 left = q_combine_ef(e.smp(S_T3), e.smp(S_T3 + 1), e.smp(S_T3 + 2), e.smp(S_T3 + 3));
        QM31 right = q_combine_ef(e.smp(S_T3 + 4), e.smp(S_T3 + 5), e.smp(S_T3 + 6), e.smp(S_T3 + 7));
        uint32_t bound = umax(lp + 2, lq + 3);
        expected_out = q_add(left, q_mul(right, q_double_x(ox, bound - 2)));
    }
    acc_out = e.acc;
    prefix_diff_out = e.prefix_diff;
}

// the u64 column-count prefixes of the four trees of sampled_values (the column prefixes: EvalCtx::smp / prefix_diff_row)
__device__ __forceinline__ uint32_t tree_prefix_diff(const uint32_t* w) {
    uint32_t diff = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) diff |= (w[SAMPLES.tree_prefix[t]] ^ tree_cols(t)) | w[SAMPLES.tree_prefix[t] + 1];
    return diff;
}
// row form: the 134 column prefixes dealt over the 16 lanes of the row (each lane's own OR; no reduction: a lane that finds
// a wrong prefix raises the flag itself)
__device__ __forceinline__ uint32_t prefix_diff_row(const uint32_t* w, uint32_t i) {
    uint32_t diff = i == 0 ? tree_prefix_diff(w) : 0u;
    // lane i: columns i, i + 16, ... (134 columns: nine rounds, the last one partly masked); straight-line, so that the
    // table words and then the prefix words are fetched together and not one round trip after the other
#pragma unroll
    for (uint32_t t = 0; t < 9; t++) {
        const uint32_t c = i + 16u * t;
        if (c < 134u) {
            const uint32_t po = SAMPLES.col_prefix[c];
            const uint32_t cnt = (c >= 114u && c < 126u && ((c - 110u) & 4u)) ? 2u : 1u;  // n_samples_of: interaction columns 4-7, 12-15
            diff |= (w[po] ^ cnt) | w[po + 1];
        }
    }
    return diff;
}
static_assert(n_samples_of(2, 4) == 2 && n_samples_of(2, 7) == 2 && n_samples_of(2, 12) == 2 && n_samples_of(2, 3) == 1 && n_samples_of(2, 8) == 1 &&
              n_samples_of(1, 4) == 1, "prefix_diff_row restates n_samples_of on the flat column index (tree 2 = columns 110..125)");

// ------------------------------------------------------------------ row form
// The same evaluation with ONE PROOF PER 16-LANE ROW (lane i = state word i of the Poseidon AIR's round function),
// for batches too small to fill the machine with one lane per proof: there k_oods' ~880 dependent QM31
// multiplications are pure latency on the critical path (transcript -> OODS -> verdict).  Here the 78 per-word
// constraints of the Poseidon component are evaluated 16 at a time; the random-coefficient accumulation
//     acc = sum_k c_k * dinv_k * rc^(85-k)            (data_structures.rs:26-28: acc = acc*rc + c*dinv, 86 times)
// is regrouped so that lane i scales its own constraints by rc^(15-i) times a per-group power, and the lanes'
// shares are added with a row all-reduce.  The external / internal matrices act coordinate-wise on QM31, so they are
// the DPP forms of poseidon2_row.hpp applied to each of the 4 coordinates.  ~5x shorter latency, ~3x more issue slots
// per proof than the lane form: the host picks by batch size (RSV_OODS=row|lane overrides, for tests).
__device__ __forceinline__ QM31 q_mds_row(QM31 x, bool odd) {
    return q_mk(mds_row(x.a.a, odd), mds_row(x.a.b, odd), mds_row(x.b.a, odd), mds_row(x.b.b, odd));
}
__device__ __forceinline__ QM31 q_sum_row(QM31 x) {
    return q_mk(sum_row(x.a.a), sum_row(x.a.b), sum_row(x.b.a), sum_row(x.b.b));
}
__device__ __forceinline__ QM31 q_sel(bool c, QM31 a, QM31 b) {
    return q_mk(c ? a.a.a : b.a.a, c ? a.a.b : b.a.b, c ? a.b.a : b.b.a, c ? a.b.b : b.b.b);
}

// Every lane of the row must be active.  The results are the same on all 16 lanes.
__device__ __noinline__ void oods_eval_row(const uint32_t* w, uint32_t lp, uint32_t lq, QM31 plonk_sum, QM31 poseidon_sum,
                                           QM31 z, QM31 alpha, QM31 rc, QM31 ox, uint32_t i, QM31& acc_out,
                                           QM31& expected_out) {
    const QM31 one = q_one();
    const bool odd = i & 1u;
    auto smp = [&](int k) { return ldq(w + SAMPLES.off[k]); };
    // pi^k(oods.x) for k = lp-1, lq-1 and bound-2 = max(lp, lq+1) from ONE doubling chain; both inverses from one
    const uint32_t kb = umax(lp, lq + 1u);
    QM31 vp = ox, vq = ox, vb = ox, cur = ox;
#pragma unroll 1
    for (uint32_t k = 1; k <= kb; k++) {
        cur = q_sub(q_dbl(q_mul(cur, cur)), one);
        if (k == lp - 1u) vp = cur;
        if (k == lq - 1u) vq = cur;
        if (k == kb) vb = cur;
    }
    // (a vanishing value of zero — an OODS point on one of the trace cosets, reachable only through the probe — has
    // the inverse 0 the lane form's separate q_inv gives it, and must not zero the other component's inverse)
    QM31 dinv_p, dinv_q;
    {
        const bool zp = q_eq(vp, q_zero()), zq = q_eq(vq, q_zero());
        const QM31 fp = zp ? one : vp, fq = zq ? one : vq;
        QM31 ti = q_inv(q_mul(fp, fq));
        dinv_p = zp ? q_zero() : q_mul(ti, fq);
        dinv_q = zq ? q_zero() : q_mul(ti, fp);
    }
    // ---- plonk component (6 constraints): sequential, identical on every lane
    EvalCtx e;
    e.rc = rc; e.acc = q_zero(); e.z = z; e.alpha = alpha; e.alpha2 = q_mul(alpha, alpha); e.w = w;
    {
        e.dinv = dinv_p;
        e.shift = q_mul_m(plonk_sum, m_inv(1u << lp));
        e.inter = S_T2;
        const int pre = S_T0, tr = S_T1;
        QM31 enforce = e.smp(pre + 9), op = e.smp(pre + 3);
        e.constraint(q_mul(enforce, e.smp(tr + 9)));
        e.constraint(q_mul(enforce, e.smp(tr + 10)));
        e.constraint(q_mul(enforce, e.smp(tr + 11)));
        QM31 a = q_combine_ef(e.smp(tr + 0), e.smp(tr + 1), e.smp(tr + 2), e.smp(tr + 3));
        QM31 b = q_combine_ef(e.smp(tr + 4), e.smp(tr + 5), e.smp(tr + 6), e.smp(tr + 7));
        QM31 cc = q_combine_ef(e.smp(tr + 8), e.smp(tr + 9), e.smp(tr + 10), e.smp(tr + 11));
        e.constraint(q_sub(q_sub(cc, q_mul(op, q_add(a, b))), q_mul(q_mul(q_sub(one, op), a), b)));
        e.relation<0>(e.smp(pre + 4), a, e.smp(pre + 0));
        e.relation<1>(e.smp(pre + 5), b, e.smp(pre + 1));
        e.relation<2>(e.smp(pre + 6), cc, e.smp(pre + 2));
        e.relation<3>(q_neg(e.smp(pre + 8)), e.smp(pre + 7), a, b);
        e.finalize_logup<4, 2>();
    }
    const QM31 acc_plonk = e.acc;
    // ---- poseidon component: 78 per-word constraints on the lanes + 2 logup constraints
    const int pre = S_T0 + 10, in = S_T1 + 12, mid = in + 16, out = in + 32;
    const int rc0 = pre + 4, rc1 = pre + 20;
    const QM31 is_first = smp(pre), is_last = smp(pre + 1), is_full = smp(pre + 2), round_id = smp(pre + 3);
    const QM31 not_first = q_sub(one, is_first), not_last = q_sub(one, is_last), is_partial = q_sub(not_first, is_full);
    const QM31 swap_val = smp(mid), one_minus_swap = q_sub(one, swap_val);
    const QM31 in_i = smp(in + (int)i), mid_i = smp(mid + (int)i), out_i = smp(out + (int)i);
    // powers of the random coefficient: lane i scales by rc^(15-i), a group by rc^{64,48,32,16,2}
    const QM31 r2 = q_mul(rc, rc), r4 = q_mul(r2, r2), r8 = q_mul(r4, r4), r16 = q_mul(r8, r8), r32 = q_mul(r16, r16),
               r64 = q_mul(r32, r32), r48 = q_mul(r32, r16), r80 = q_mul(r64, r16);
    QM31 ri;
    {
        const uint32_t ex = 15u - i;
        ri = q_sel(ex & 1u, rc, one);
        ri = q_mul(ri, q_sel(ex & 2u, r2, one));
        ri = q_mul(ri, q_sel(ex & 4u, r4, one));
        ri = q_mul(ri, q_sel(ex & 8u, r8, one));
    }
    QM31 share;  // sum over this lane's constraints of c * rc^(group power), before the common rc^(15-i)
    {   // first round: swap-mix of the two halves, external matrix (poseidon.rs:122-140)
        const QM31 lo = smp(in + (int)(i & 7u)), hi = smp(in + (int)(i & 7u) + 8);
        QM31 st = i < 8 ? q_add(q_mul(lo, one_minus_swap), q_mul(hi, swap_val))
                        : q_add(q_mul(lo, swap_val), q_mul(hi, one_minus_swap));
        st = q_mds_row(st, odd);
        share = q_mul(q_mul(is_first, q_sub(st, out_i)), r64);
    }
    {   // full rounds (poseidon.rs:142-170)
        const QM31 full = q_pow5(q_add(in_i, smp(rc0 + (int)i)));
        share = q_add(share, q_mul(q_mul(is_full, q_sub(mid_i, full)), r48));
        QM31 st = q_mds_row(mid_i, odd);
        st = q_pow5(q_add(st, smp(rc1 + (int)i)));
        st = q_mds_row(st, odd);
        share = q_add(share, q_mul(q_mul(is_full, q_sub(out_i, st)), r32));
    }
    {   // partial rounds (poseidon.rs:172-196).  The chain continues from the SAMPLED mid value of each round, so the
        // 14 S-boxes do not depend on each other: word 0 before round r is known to every lane (3*mid[r-1] + row sum),
        // lane r keeps it and raises it to the fifth power afterwards, all 14 at once.
        QM31 st = in_i, st0 = smp(in), mine = q_zero();
#pragma unroll 1
        for (int r = 0; r < 14; r++) {
            mine = q_sel(i == (uint32_t)r, st0, mine);
            const QM31 mr = smp(mid + r);
            st = q_sel(i == 0u, mr, st);
            const QM31 sum = q_sum_row(st);
            const uint32_t dg = i == 0u ? 3u : (1u << (i + 1u));
            st = q_add(sum, q_mul_m(st, dg));
            st0 = q_add(sum, q_mul_m(mr, 3u));
        }
        const int rr = (int)(i < 14u ? i : 13u);
        const QM31 v = q_pow5(q_add(mine, smp(rc0 + rr)));
        QM31 cm = q_mul(is_partial, q_sub(smp(mid + rr), v));
        cm = q_sel(i < 14u, cm, q_zero());
        share = q_add(share, q_mul(cm, r16));
        share = q_add(share, q_mul(q_mul(is_partial, q_sub(out_i, st)), r2));
    }
    const QM31 lanes = q_sum_row(q_mul(share, ri));
    // logup of the poseidon component (poseidon.rs:198-239): c_78 * rc + c_79, identical on every lane
    EvalCtx g;
    g.rc = rc; g.acc = q_zero(); g.z = z; g.alpha = alpha; g.alpha2 = e.alpha2; g.w = w; g.dinv = one;
    g.shift = q_mul_m(poseidon_sum, m_inv(1u << lq));
    g.inter = S_T2 + 12;
    {
        QM31 ext1 = smp(pre + 36), ext2 = smp(pre + 37), ext1_nz = smp(pre + 38), ext2_nz = smp(pre + 39);
        QM31 in_left = q_dbl(round_id), in_right = q_add(in_left, one), out_left = q_add(in_right, one),
             out_right = q_add(out_left, one);
#define EF4(base) q_combine_ef(smp(base), smp((base) + 1), smp((base) + 2), smp((base) + 3))
        g.relation<0>(q_sub(q_mul(ext1_nz, is_first), not_first), q_add(q_mul(is_first, ext1), q_mul(not_first, in_left)),
                   EF4(in), EF4(in + 4));
        g.relation<1>(q_sub(q_mul(ext2_nz, is_first), not_first), q_add(q_mul(is_first, ext2), q_mul(not_first, in_right)),
                   EF4(in + 8), EF4(in + 12));
        g.relation<2>(q_add(q_mul(ext1_nz, is_last), not_last), q_add(q_mul(is_last, ext1), q_mul(not_last, out_left)),
                   EF4(out), EF4(out + 4));
        g.relation<3>(q_add(q_mul(ext2_nz, is_last), not_last), q_add(q_mul(is_last, ext2), q_mul(not_last, out_right)),
                   EF4(out + 8), EF4(out + 12));
#undef EF4
        g.relation<4>(q_mul(is_first, not_last), swap_val, smp(rc0));
        g.finalize_logup<5, 3>();
    }
    acc_out = q_add(q_mul(acc_plonk, r80), q_mul(dinv_q, q_add(lanes, g.acc)));
    {   // composition/src/lib.rs:106-120
        QM31 left = q_combine_ef(smp(S_T3), smp(S_T3 + 1), smp(S_T3 + 2), smp(S_T3 + 3));
        QM31 right = q_combine_ef(smp(S_T3 + 4), smp(S_T3 + 5), smp(S_T3 + 6), smp(S_T3 + 7));
        expected_out = q_add(left, q_mul(right, vb));
    }
}

// 4 waves per SIMD asked for (<= 128 registers, the rest spilled to scratch): the kernel is one latency-bound wave per
// SIMD that lives for milliseconds underneath the Merkle kernels, and every 128 registers it holds cost that SIMD a
// Merkle wave for as long.
// (255 registers, one wave per SIMD: neither a launch bound nor amdgpu_waves_per_eu brings it down without spilling what the
// LDS state just saved; beside the FRI trees — whose waves leave a SIMD 32 registers free — its waves start as those retire)
__global__ __launch_bounds__(64, 4) void k_oods(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
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
    QM31 acc, expected;
    __shared__ uint32_t qstate[QSTATE_WORDS];
    uint32_t pdiff = 0;
    oods_eval(w, m.lp, m.lq, plonk_sum, poseidon_sum, z, alpha, ldq(c.rc), ldq(c.oods_x), QState{qstate + threadIdx.x}, acc, expected, pdiff);
    if (!q_eq(acc, expected)) flags |= 1u << R_COMPOSITION;
    if (pdiff | tree_prefix_diff(w)) flags |= 1u << R_PARSE;  // bincode shape of sampled_values (k_parse.hpp)
    if (flags) atomicOr(&c.flags, flags);
}

// Row-form pipeline kernel: 16 lanes per proof; whole rows leave together (DPP needs every lane of a live row).
__global__ __launch_bounds__(256) void k_oods_row(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets,
                                                  uint32_t n, const ProofMeta* __restrict__ metas,
                                                  ProofCtx* __restrict__ ctxs, const PubInput* __restrict__ pi,
                                                  uint32_t n_pi) {
    const uint32_t i = threadIdx.x & 15u;
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= n) return;
    const ProofMeta& m = metas[p];
    if (m.reason != R_OK) return;
    ProofCtx& c = ctxs[p];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    uint32_t flags = 0;
    const uint32_t pdiff = prefix_diff_row(w, i);  // (first: its loads are in flight underneath the arithmetic below)
    QM31 z = ldq(c.z), alpha = ldq(c.alpha), plonk_sum = ldq(w + W_PLONK_SUM), poseidon_sum = ldq(w + W_POSEIDON_SUM);
    {  // fiat_shamir/src/lib.rs:133-141: the public inputs are dealt to the lanes, their fractions added over the row
        QM31 sum = q_zero();
        for (uint32_t k = i; k < n_pi; k += 16u) {
            QM31 dnm = q_sub(q_add(ldq(pi[k].value), q_mul_m(alpha, pi[k].idx % P)), z);
            sum = q_add(sum, q_inv(dnm));
        }
        sum = q_sum_row(sum);
        if (!q_eq(q_add(q_add(sum, poseidon_sum), plonk_sum), q_zero())) flags |= 1u << R_LOGUP;
    }
    QM31 acc, expected;
    oods_eval_row(w, m.lp, m.lq, plonk_sum, poseidon_sum, z, alpha, ldq(c.rc), ldq(c.oods_x), i, acc, expected);
    if (!q_eq(acc, expected)) flags |= 1u << R_COMPOSITION;
    if (flags && i == 0) atomicOr(&c.flags, flags);
    if (pdiff) atomicOr(&c.flags, 1u << R_PARSE);  // bincode shape of sampled_values (k_parse.hpp)
}

__global__ __launch_bounds__(256) void k_oods_row_probe(const uint32_t* __restrict__ prefix, uint32_t prefix_words,
                                                        const uint32_t* __restrict__ params, uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = threadIdx.x & 15u;
    const uint32_t it = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (it >= n) return;
    const uint32_t* pr = params + (size_t)it * 26;
    QM31 acc, expected;
    oods_eval_row(prefix + (size_t)it * prefix_words, pr[0], pr[1], ldq(pr + 2), ldq(pr + 6), ldq(pr + 10), ldq(pr + 14),
                  ldq(pr + 18), ldq(pr + 22), i, acc, expected);
    if (i == 5) {  // any lane: the results are row-uniform
        stq(out + (size_t)it * 8, acc);
        stq(out + (size_t)it * 8 + 4, expected);
    }
}

// rsv_transcript (the one-proof probe runs the parser and the transcript only): the prefixes of sampled_values, one lane per proof
__global__ __launch_bounds__(64) void k_prefix_check(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets, uint32_t n,
                                                     const ProofMeta* __restrict__ metas, ProofCtx* __restrict__ ctxs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || metas[p].reason != R_OK) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + offsets[p]);
    uint32_t diff = tree_prefix_diff(w);
    for (uint32_t k = 0; k < (uint32_t)N_SAMPLES; k++) {
        const uint32_t po = SAMPLES.pre_of[k];
        if (po) diff |= (w[po] ^ (uint32_t)SAMPLES.cnt_of[k]) | w[po + 1];
    }
    if (diff) atomicOr(&ctxs[p].flags, 1u << R_PARSE);
}

// Probe (rsv_oods_eval): item i = a proof prefix of OODS_PREFIX_WORDS words (only the sampled values are read) plus
// 26 parameter words lp, lq, plonk_sum, poseidon_sum, z, alpha, random_coeff, oods.x; out = accumulator | expected.
constexpr uint32_t OODS_PARAM_WORDS = 26;
__global__ __launch_bounds__(64, 4) void k_oods_probe(const uint32_t* __restrict__ prefix, uint32_t prefix_words,
                                                   const uint32_t* __restrict__ params, uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* pr = params + (size_t)i * OODS_PARAM_WORDS;
    QM31 acc, expected;
    __shared__ uint32_t qstate[QSTATE_WORDS];
    uint32_t pdiff = 0;  // (the probe evaluates on arbitrary samples: the prefixes are the full pipeline's business)
    oods_eval(prefix + (size_t)i * prefix_words, pr[0], pr[1], ldq(pr + 2), ldq(pr + 6), ldq(pr + 10), ldq(pr + 14), ldq(pr + 18),
              ldq(pr + 22), QState{qstate + threadIdx.x}, acc, expected, pdiff);
    stq(out + (size_t)i * 8, acc);
    stq(out + (size_t)i * 8 + 4, expected);
}

}  // namespace rsv
