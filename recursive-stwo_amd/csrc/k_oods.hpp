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

}  // namespace rsv
