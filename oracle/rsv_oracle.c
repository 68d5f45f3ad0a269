/*
 * rsv_oracle.c — TEST INFRASTRUCTURE ONLY (see rsv_oracle.h).
 *
 * CPU restatement, in plain C, of the values side of recursive-stwo's verify
 * path.  It follows the in-tree Rust sources function by function (citations
 * are `path:line` relative to the reference root) plus the stwo-resident
 * conventions listed in SURVEY.md Appendix B.  Merkle decommitments are
 * checked in stwo's *batched* form (one walk per tree, shared nodes hashed
 * once) — deliberately a different formulation from the product's
 * lane-per-path kernels so the two cross-check each other.
 *
 * Parity: pinned by the Poseidon2 KAT and by acceptance of the reference's 15
 * Poseidon-channel fixtures (tests/test_oracle.py).
 */
#include "rsv_oracle.h"

#include <stdlib.h>
#include <string.h>

#define P RSV_M31_P
#define MAX_QUERIES 128
#define MAX_LAYERS 32
#define N_SAMPLES 142
#define N_COLS_TOTAL 134

typedef uint32_t m31;
typedef struct { m31 a, b; } cm31;  /* a + b*i,  i^2 = -1 */
typedef struct { cm31 a, b; } qm31; /* a + b*u,  u^2 = 2 + i */

/* ------------------------------------------------------------------ a1: M31
 * stwo M31, used through primitives/fields/src/m31.rs:62-115,140-156. */
static inline m31 m_add(m31 x, m31 y) { uint32_t s = x + y; return s >= P ? s - P : s; }
static inline m31 m_sub(m31 x, m31 y) { return x >= y ? x - y : x + P - y; }
static inline m31 m_neg(m31 x) { return x ? P - x : 0; }
/* Mersenne fold (2^31 = 1 mod P) instead of a division: what any tuned CPU implementation does, so that the CPU baseline
 * timed from this file is not handicapped by `% P`.  Correct for ANY 32-bit operands (words >= P can reach here only
 * on paths that reject them afterwards): t < 2^64, first fold < 2^34, second fold <= P + 7. */
static inline m31 m_mul(m31 x, m31 y) {
    uint64_t t = (uint64_t)x * y;
    uint64_t f = (t & P) + (t >> 31);
    uint32_t r = (uint32_t)(f & P) + (uint32_t)(f >> 31);
    return r >= P ? r - P : r;
}
static m31 m_pow(m31 x, uint32_t e) {
    m31 r = 1;
    while (e) { if (e & 1) r = m_mul(r, x); x = m_mul(x, x); e >>= 1; }
    return r;
}
static inline m31 m_inv(m31 x) { return m_pow(x, P - 2); }

/* --------------------------------------------------------- a2: CM31 / QM31
 * primitives/fields/src/cm31.rs:87-192, primitives/fields/src/qm31.rs:87-249,360-366,402-418 */
static inline cm31 c_mk(m31 a, m31 b) { cm31 r = {a, b}; return r; }
static inline cm31 c_add(cm31 x, cm31 y) { return c_mk(m_add(x.a, y.a), m_add(x.b, y.b)); }
static inline cm31 c_sub(cm31 x, cm31 y) { return c_mk(m_sub(x.a, y.a), m_sub(x.b, y.b)); }
static inline cm31 c_mul(cm31 x, cm31 y) {
    return c_mk(m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a)));
}
static inline cm31 c_mul_m(cm31 x, m31 k) { return c_mk(m_mul(x.a, k), m_mul(x.b, k)); }
static inline cm31 c_inv(cm31 x) {
    m31 n = m_inv(m_add(m_mul(x.a, x.a), m_mul(x.b, x.b)));
    return c_mk(m_mul(x.a, n), m_mul(m_neg(x.b), n));
}
static inline qm31 q_mk(m31 a0, m31 a1, m31 b0, m31 b1) { qm31 r = {{a0, a1}, {b0, b1}}; return r; }
static inline qm31 q_from_m(m31 x) { return q_mk(x, 0, 0, 0); }
static const qm31 Q_ZERO = {{0, 0}, {0, 0}};
static const qm31 Q_ONE = {{1, 0}, {0, 0}};
static inline qm31 q_add(qm31 x, qm31 y) { qm31 r = {c_add(x.a, y.a), c_add(x.b, y.b)}; return r; }
static inline qm31 q_sub(qm31 x, qm31 y) { qm31 r = {c_sub(x.a, y.a), c_sub(x.b, y.b)}; return r; }
static inline qm31 q_neg(qm31 x) { return q_sub(Q_ZERO, x); }
static inline qm31 q_mul(qm31 x, qm31 y) {
    /* (a + bu)(c + du) = ac + bd(2+i) + (ad + bc)u */
    cm31 ac = c_mul(x.a, y.a), bd = c_mul(x.b, y.b);
    cm31 bd_r = c_mk(m_sub(m_add(bd.a, bd.a), bd.b), m_add(m_add(bd.b, bd.b), bd.a)); /* bd*(2+i) */
    qm31 r = {c_add(ac, bd_r), c_add(c_mul(x.a, y.b), c_mul(x.b, y.a))};
    return r;
}
static inline qm31 q_mul_m(qm31 x, m31 k) { qm31 r = {c_mul_m(x.a, k), c_mul_m(x.b, k)}; return r; }
static inline qm31 q_mul_c(qm31 x, cm31 k) { qm31 r = {c_mul(x.a, k), c_mul(x.b, k)}; return r; }
static inline qm31 q_inv(qm31 x) {
    /* 1/(a+bu) = (a-bu)/(a^2 - (2+i) b^2) */
    cm31 b2 = c_mul(x.b, x.b);
    cm31 b2r = c_mk(m_sub(m_add(b2.a, b2.a), b2.b), m_add(m_add(b2.b, b2.b), b2.a));
    cm31 den = c_inv(c_sub(c_mul(x.a, x.a), b2r));
    qm31 r = {c_mul(x.a, den), c_mul(c_sub(c_mk(0, 0), x.b), den)};
    return r;
}
static inline int q_eq(qm31 x, qm31 y) {
    return x.a.a == y.a.a && x.a.b == y.a.b && x.b.a == y.b.a && x.b.b == y.b.b;
}
/* shift_by_i / shift_by_j / combine_ef: primitives/fields/src/qm31.rs:402-418,466-468;
 * components/recursive/composition/src/data_structures.rs:142-145 */
static inline qm31 q_combine_ef(qm31 v0, qm31 v1, qm31 v2, qm31 v3) {
    const qm31 I = {{0, 1}, {0, 0}}, U = {{0, 0}, {1, 0}}, IU = {{0, 0}, {0, 1}};
    return q_add(q_add(v0, q_mul(v1, I)), q_add(q_mul(v2, U), q_mul(v3, IU)));
}

void rsvo_qm31_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    qm31 r = q_mul(q_mk(a[0], a[1], a[2], a[3]), q_mk(b[0], b[1], b[2], b[3]));
    out[0] = r.a.a; out[1] = r.a.b; out[2] = r.b.a; out[3] = r.b.b;
}
void rsvo_qm31_inv(const uint32_t* a, uint32_t* out) {
    qm31 r = q_inv(q_mk(a[0], a[1], a[2], a[3]));
    out[0] = r.a.a; out[1] = r.a.b; out[2] = r.b.a; out[3] = r.b.b;
}

/* ------------------------------------------------------------ a3: Poseidon2
 * primitives/poseidon31/src/implementation.rs:7-149; constants are the
 * Poseidon2 parameters for p = 2^31-1, t = 16 listed in
 * primitives/poseidon31/src/parameters.rs:6-190 (partial-round diagonal
 * 3,4,8,...,65536 is generated below). */
static const m31 RC_FIRST[4][16] = {
    {0x768bab52, 0x70e0ab7d, 0x3d266c8a, 0x6da42045, 0x600fef22, 0x41dace6b, 0x64f9bdd4, 0x5d42d4fe,
     0x76b1516d, 0x6fc9a717, 0x70ac4fb6, 0x00194ef6, 0x22b644e2, 0x1f7916d5, 0x47581be2, 0x2710a123},
    {0x6284e867, 0x018d3afe, 0x5df99ef3, 0x4c1e467b, 0x566f6abc, 0x2994e427, 0x538a6d42, 0x5d7bf2cf,
     0x7fda2dab, 0x0fd854c4, 0x46922fca, 0x3d7763a1, 0x19fd05ca, 0x0a4bbb43, 0x15075851, 0x3d903d76},
    {0x2d290ff7, 0x40809fa0, 0x59dac6ec, 0x127927a2, 0x6bbf0ea0, 0x0294140f, 0x24742976, 0x6e84c081,
     0x22484f4a, 0x354cae59, 0x0453ffe1, 0x3f47a3cc, 0x0088204e, 0x6066e109, 0x3b7c4b80, 0x6b55665d},
    {0x3bc4b897, 0x735bf378, 0x508daf42, 0x1884fc2b, 0x7214f24c, 0x7498be0a, 0x1a60e640, 0x3303f928,
     0x29b46376, 0x5c96bb68, 0x65d097a5, 0x1d358e9f, 0x4a9a9017, 0x4724cf76, 0x347af70f, 0x1e77e59a}};
static const m31 RC_PARTIAL[14] = {0x7f7ec4bf, 0x0421926f, 0x5198e669, 0x34db3148, 0x4368bafd,
                                   0x66685c7f, 0x78d3249a, 0x60187881, 0x76dad67a, 0x0690b437,
                                   0x1ea95311, 0x40e5369a, 0x38f103fc, 0x1d226a21};
static const m31 RC_LAST[4][16] = {
    {0x57090613, 0x1fa42108, 0x17bbef50, 0x1ff7e11c, 0x047b24ca, 0x4e140275, 0x4fa086f5, 0x079b309c,
     0x1159bd47, 0x6d37e4e5, 0x075d8dce, 0x12121ca0, 0x7f6a7c40, 0x68e182ba, 0x5493201b, 0x0444a80e},
    {0x0064f4c6, 0x6467abe6, 0x66975762, 0x2af68f9b, 0x345b33be, 0x1b70d47f, 0x053db717, 0x381189cb,
     0x43b915f8, 0x20df3694, 0x0f459d26, 0x77a0e97b, 0x2f73e739, 0x1876c2f9, 0x65a0e29a, 0x4cabefbe},
    {0x5abd1268, 0x4d34a760, 0x12771799, 0x69a0c9ac, 0x39091e55, 0x7f611cd0, 0x3af055da, 0x7ac0bbdf,
     0x6e0f3a24, 0x41e3b6f7, 0x49b3756d, 0x568bc538, 0x20c079d8, 0x1701c72c, 0x7670dc6c, 0x5a439035},
    {0x7c93e00e, 0x561fbb4d, 0x1178907b, 0x02737406, 0x32fb24f1, 0x6323b60a, 0x6ab12418, 0x42c99cea,
     0x155a0b97, 0x53d1c6aa, 0x2bd20347, 0x279b3d73, 0x4f5f3c70, 0x0245af6c, 0x238359d3, 0x49966a59}};

/* the tables above, for rsv_emulated.c (0: first four rounds [4][16], 1: partial [14], 2: last four [4][16]) */
const uint32_t* rsvo_round_constants(int which) {
    return which == 0 ? &RC_FIRST[0][0] : which == 1 ? RC_PARTIAL : which == 2 ? &RC_LAST[0][0] : NULL;
}

static __thread uint64_t g_perm_count;
uint64_t rsvo_perm_count(void) { return g_perm_count; }
void rsvo_perm_count_reset(void) { g_perm_count = 0; }

/* implementation.rs:7-19 */
static void mds4(m31* x) {
    m31 t0 = m_add(x[0], x[1]), t1 = m_add(x[2], x[3]);
    m31 t2 = m_add(m_add(x[1], x[1]), t1), t3 = m_add(m_add(x[3], x[3]), t0);
    m31 t1_4 = m_add(m_add(t1, t1), m_add(t1, t1)), t0_4 = m_add(m_add(t0, t0), m_add(t0, t0));
    m31 t4 = m_add(t1_4, t3), t5 = m_add(t0_4, t2);
    x[0] = m_add(t3, t5); x[1] = t5; x[2] = m_add(t2, t4); x[3] = t4;
}
/* implementation.rs:21-58: circ(2*M4, M4, M4, M4) */
static void mds16(m31* s) {
    for (int g = 0; g < 4; g++) mds4(s + 4 * g);
    for (int j = 0; j < 4; j++) {
        m31 sum = m_add(m_add(s[j], s[j + 4]), m_add(s[j + 8], s[j + 12]));
        for (int g = 0; g < 4; g++) s[4 * g + j] = m_add(s[4 * g + j], sum);
    }
}
static inline m31 pow5(m31 x) { m31 x2 = m_mul(x, x); return m_mul(m_mul(x2, x2), x); }

/* implementation.rs:108-149 */
static void poseidon2(m31* s) {
    g_perm_count++;
    mds16(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow5(m_add(s[i], RC_FIRST[r][i]));
        mds16(s);
    }
    for (int r = 0; r < 14; r++) {
        s[0] = pow5(m_add(s[0], RC_PARTIAL[r]));
        m31 sum = 0;
        for (int i = 0; i < 16; i++) sum = m_add(sum, s[i]);
        s[0] = m_add(sum, m_mul(s[0], 3));
        for (int i = 1; i < 16; i++) s[i] = m_add(sum, m_mul(s[i], 1u << (i + 1)));
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = pow5(m_add(s[i], RC_LAST[r][i]));
        mds16(s);
    }
}

static int canonical(const uint32_t* w, size_t n) {
    for (size_t i = 0; i < n; i++) if (w[i] >= P) return 0;
    return 1;
}

int rsvo_poseidon2_permute(const uint32_t* in16, uint32_t* out16, size_t n) {
    if (n && (!in16 || !out16)) return RSV_E_NULL;
    if (!canonical(in16, 16 * n)) return RSV_E_RANGE;
    for (size_t i = 0; i < n; i++) {
        m31 s[16];
        memcpy(s, in16 + 16 * i, 64);
        poseidon2(s);
        memcpy(out16 + 16 * i, s, 64);
    }
    return RSV_OK;
}

/* a4: primitives/poseidon31/src/lib.rs:282-311 (state = left||right, or right||left on swap) */
/* Optional recorder (rsvo_poseidon_flow): every invocation appends what PoseidonFlow holds for it
 * (constraint_system/src/plonk_with_poseidon.rs:117-128, primitives/poseidon31/src/lib.rs:385-415): the two input
 * halves AS GIVEN (the accelerator applies the swap), the two output halves, the swap bit. */
typedef struct { uint32_t* rec; size_t cap, n; } flow_recorder;
static __thread flow_recorder* g_flow;
static void half_permute(const m31* left, const m31* right, int swap, m31* rate, m31* cap) {
    m31 s[16];
    memcpy(s, swap ? right : left, 32);
    memcpy(s + 8, swap ? left : right, 32);
    poseidon2(s);
    if (g_flow) {
        if (g_flow->n < g_flow->cap) {
            uint32_t* r = g_flow->rec + 33 * g_flow->n;
            memcpy(r, left, 32); memcpy(r + 8, right, 32); memcpy(r + 16, s, 64); r[32] = swap ? 1u : 0u;
        }
        g_flow->n++;
    }
    if (rate) memcpy(rate, s, 32);
    if (cap) memcpy(cap, s + 8, 32);
}
int rsvo_poseidon2_half_permute(const uint32_t* left8, const uint32_t* right8, const uint8_t* swap,
                                uint32_t* out_rate8, uint32_t* out_cap8, size_t n) {
    if (n && (!left8 || !right8)) return RSV_E_NULL;
    if (!canonical(left8, 8 * n) || !canonical(right8, 8 * n)) return RSV_E_RANGE;
    for (size_t i = 0; i < n; i++)
        half_permute(left8 + 8 * i, right8 + 8 * i, swap ? swap[i] : 0,
                     out_rate8 ? out_rate8 + 8 * i : NULL, out_cap8 ? out_cap8 + 8 * i : NULL);
    return RSV_OK;
}

/* ------------------------------------------------------- a5: Merkle hasher
 * primitives/merkle/src/lib.rs:141-181 (hash_m31_columns_get_capacity):
 * d = 0; for each zero-padded 8-word chunk: d = perm(chunk || d)[8..16] */
static void sponge_capacity(const m31* cols, size_t n, m31* d) {
    static const m31 zero8[8] = {0};
    memset(d, 0, 32);
    for (size_t off = 0; off < n; off += 8) {
        m31 chunk[8] = {0};
        size_t k = n - off < 8 ? n - off : 8;
        memcpy(chunk, cols + off, 4 * k);
        half_permute(chunk, d, 0, NULL, d);
    }
    (void)zero8;
}
/* stwo Poseidon31MerkleHasher::hash_node == primitives/merkle/src/lib.rs:9-20,50-91:
 *   leaf            = perm(0 || sponge(cols))[0..8]
 *   node            = perm(L || R)[0..8]
 *   node + columns  = perm(perm(L||R)[0..8] || sponge(cols))[0..8] */
static void hash_node(const m31* left, const m31* right, const m31* cols, size_t n_cols, m31* out) {
    m31 d[8];
    if (!left) {
        static const m31 zero8[8] = {0};
        sponge_capacity(cols, n_cols, d);
        half_permute(zero8, d, 0, out, NULL);
        return;
    }
    half_permute(left, right, 0, out, NULL);
    if (n_cols) {
        sponge_capacity(cols, n_cols, d);
        half_permute(out, d, 0, out, NULL);
    }
}
int rsvo_merkle_hash_node(const uint32_t* left8, const uint32_t* right8, const uint32_t* cols,
                          size_t n_cols, uint32_t* out8, size_t n) {
    if (n && !out8) return RSV_E_NULL;
    if ((left8 == NULL) != (right8 == NULL)) return RSV_E_NULL;
    if (n_cols && !cols) return RSV_E_NULL;
    if (!left8 && n_cols == 0) return RSV_E_SIZE;
    if (left8 && (!canonical(left8, 8 * n) || !canonical(right8, 8 * n))) return RSV_E_RANGE;
    if (!canonical(cols, n_cols * n)) return RSV_E_RANGE;
    for (size_t i = 0; i < n; i++)
        hash_node(left8 ? left8 + 8 * i : NULL, right8 ? right8 + 8 * i : NULL,
                  cols ? cols + n_cols * i : NULL, n_cols, out8 + 8 * i);
    return RSV_OK;
}

/* a9: SinglePathMerkleProof::verify (components/hints/src/decommit.rs:22-42) */
int rsvo_merkle_path_root(const uint32_t* query, const uint32_t* sib8, const uint32_t* cols,
                          const uint32_t* n_cols_at, uint32_t depth, uint32_t* out_root8, size_t n) {
    if (!query || !n_cols_at || !out_root8 || (depth && !sib8)) return RSV_E_NULL;
    if (depth > 31 || n_cols_at[depth] == 0) return RSV_E_SIZE;
    size_t per_path = 0;
    for (uint32_t h = 0; h <= depth; h++) per_path += n_cols_at[h];
    if (!cols) return RSV_E_NULL;
    for (size_t i = 0; i < n; i++) {
        const m31* c = cols + per_path * i;
        m31 cur[8];
        hash_node(NULL, NULL, c, n_cols_at[depth], cur);
        c += n_cols_at[depth];
        for (uint32_t lvl = 0; lvl < depth; lvl++) {
            uint32_t h = depth - lvl - 1;
            const m31* sib = sib8 + 8 * ((size_t)depth * i + lvl);
            m31 nxt[8];
            if ((query[i] >> lvl) & 1) hash_node(sib, cur, c, n_cols_at[h], nxt);
            else hash_node(cur, sib, c, n_cols_at[h], nxt);
            c += n_cols_at[h];
            memcpy(cur, nxt, 32);
        }
        memcpy(out_root8 + 8 * i, cur, 32);
    }
    return RSV_OK;
}

/* ----------------------------------------------------------- a6: channel
 * primitives/channel/src/lib.rs:24-58 */
typedef struct { m31 digest[8]; uint32_t n_sent; } channel;
static void ch_mix8(channel* c, const m31* left8) { half_permute(left8, c->digest, 0, NULL, c->digest); c->n_sent = 0; }
static void ch_mix_root(channel* c, const m31* root) { ch_mix8(c, root); }
static void ch_mix_two(channel* c, qm31 f, qm31 g) {
    m31 l[8] = {f.a.a, f.a.b, f.b.a, f.b.b, g.a.a, g.a.b, g.b.a, g.b.b};
    ch_mix8(c, l);
}
static void ch_mix_one(channel* c, qm31 f) { ch_mix_two(c, f, Q_ZERO); }
static void ch_draw(channel* c, qm31* a, qm31* b) {
    m31 l[8] = {c->n_sent, 0, 0, 0, 0, 0, 0, 0}, out[8];
    c->n_sent++;
    half_permute(l, c->digest, 0, out, NULL);
    *a = q_mk(out[0], out[1], out[2], out[3]);
    *b = q_mk(out[4], out[5], out[6], out[7]);
}

/* ------------------------------------------------------------ circle group
 * x^2+y^2=1 over M31, generator (2, 1268011823) of order 2^31. */
typedef struct { m31 x, y; } cpoint;
typedef struct { qm31 x, y; } qpoint;
static cpoint cp_add(cpoint p, cpoint q) {
    cpoint r = {m_sub(m_mul(p.x, q.x), m_mul(p.y, q.y)), m_add(m_mul(p.x, q.y), m_mul(p.y, q.x))};
    return r;
}
/* k * GEN for k mod 2^31 */
static cpoint cp_gen_mul(uint32_t k) {
    cpoint acc = {1, 0}, g = {2, 1268011823u};
    for (int i = 0; i < 31; i++) {
        if ((k >> i) & 1) acc = cp_add(acc, g);
        g = cp_add(g, g);
    }
    return acc;
}
/* g_n = GEN * 2^(31-n), order 2^n */
static cpoint cp_subgroup_gen(uint32_t n) { return cp_gen_mul(n == 0 ? 0 : 1u << (31 - n)); }
/* Coset::half_odds(n).at(i): initial g_{n+2}, step g_n */
static cpoint half_odds_at(uint32_t n, uint32_t i) {
    uint32_t k = (1u << (29 - n)) + (uint32_t)(((uint64_t)i << (31 - n)) & P);
    return cp_gen_mul(k & P);
}
static uint32_t bit_reverse(uint32_t v, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
/* CanonicCoset(log).circle_domain().at(bit_reverse(q, log)) — what
 * PointCarryingQueryVar::get_next_point (primitives/query/src/lib.rs:139-143) yields. */
static cpoint domain_point(uint32_t log_size, uint32_t q) {
    uint32_t i = bit_reverse(q, log_size), half = 1u << (log_size - 1);
    if (i < half) return half_odds_at(log_size - 1, i);
    cpoint p = half_odds_at(log_size - 1, i - half);
    p.y = m_neg(p.y);
    return p;
}
void rsvo_domain_point(uint32_t log_size, uint32_t q, uint32_t* xy) {
    cpoint p = domain_point(log_size, q);
    xy[0] = p.x; xy[1] = p.y;
}
/* CirclePointQM31Var + CirclePoint<M31> (primitives/circle/src/lib.rs:236-250) */
static qpoint qp_add_m(qpoint p, cpoint q) {
    qpoint r = {q_sub(q_mul_m(p.x, q.x), q_mul_m(p.y, q.y)), q_add(q_mul_m(p.x, q.y), q_mul_m(p.y, q.x))};
    return r;
}
/* repeated_double_x_only (primitives/circle/src/lib.rs:226-233) */
static qm31 q_double_x(qm31 x, uint32_t times) {
    for (uint32_t i = 0; i < times; i++) {
        qm31 sq = q_mul(x, x);
        x = q_sub(q_add(sq, sq), Q_ONE);
    }
    return x;
}

/* ------------------------------------------------------ a13: wire format
 * bincode 1.3 of PlonkWithPoseidonProof<Poseidon31MerkleHasher>. */
typedef struct {
    const uint32_t* hash_witness; uint64_t n_hash;
} decommit_view;
typedef struct {
    const uint32_t* fri_witness; uint64_t n_witness; /* QM31 count */
    decommit_view decommit;
    const uint32_t* commitment;
} fri_layer_view;
typedef struct {
    uint32_t lp, lq;
    qm31 plonk_sum, poseidon_sum;
    rsv_pcs_config cfg;
    const uint32_t* commitments[4];
    qm31 samples[N_SAMPLES]; /* tree-major, column-major, sample-minor */
    decommit_view decommit[4];
    const uint32_t* queried[4]; uint64_t n_queried[4];
    uint64_t pow_nonce;
    fri_layer_view first;
    uint32_t n_inner;
    fri_layer_view inner[MAX_LAYERS];
    const uint32_t* last_coeffs; uint64_t n_last; uint32_t last_log_size;
    /* derived */
    uint32_t A, B, M;
} proof_view;

typedef struct { const uint8_t* p; size_t len, pos; int ok; } reader;
static uint32_t rd_u32(reader* r) {
    uint32_t v = 0;
    if (!r->ok || r->len - r->pos < 4) { r->ok = 0; return 0; }
    memcpy(&v, r->p + r->pos, 4); r->pos += 4; return v;
}
static uint64_t rd_u64(reader* r) { uint64_t lo = rd_u32(r), hi = rd_u32(r); return lo | (hi << 32); }
/* returns pointer to n u32 words (must all be canonical M31) */
static const uint32_t* rd_words(reader* r, uint64_t n) {
    if (!r->ok || n > (r->len - r->pos) / 4) { r->ok = 0; return NULL; }
    const uint32_t* w = (const uint32_t*)(r->p + r->pos);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t v; memcpy(&v, r->p + r->pos + 4 * i, 4);
        if (v >= P) { r->ok = 0; return NULL; }
    }
    r->pos += 4 * n;
    return w;
}
static qm31 rd_qm31(reader* r) {
    const uint32_t* w = rd_words(r, 4);
    return w ? q_mk(w[0], w[1], w[2], w[3]) : Q_ZERO;
}
static void rd_decommit(reader* r, decommit_view* d) {
    d->n_hash = rd_u64(r);
    if (d->n_hash > (1u << 20)) r->ok = 0;
    d->hash_witness = rd_words(r, d->n_hash * 8);
    if (rd_u64(r) != 0) r->ok = 0; /* column_witness must be empty: components/hints/src/decommit.rs:71 */
}
static void rd_fri_layer(reader* r, fri_layer_view* l) {
    l->n_witness = rd_u64(r);
    if (l->n_witness > (1u << 20)) r->ok = 0;
    l->fri_witness = rd_words(r, l->n_witness * 4);
    rd_decommit(r, &l->decommit);
    l->commitment = rd_words(r, 8);
}
static inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

static const uint32_t TREE_COLS[4] = {50, 60, 16, 8};
/* mask sizes: 1 everywhere except interaction columns 4-7 and 12-15 */
static int n_samples_of(int tree, int col) { return (tree == 2 && (col & 4)) ? 2 : 1; }

static int parse_proof(const uint8_t* bytes, size_t len, const rsv_pcs_config* want, proof_view* v) {
    reader r = {bytes, len, 0, 1};
    memset(v, 0, sizeof *v);
    if (((uintptr_t)bytes & 3) || (len & 3)) return 0;
    v->lp = rd_u32(&r); v->lq = rd_u32(&r);
    v->plonk_sum = rd_qm31(&r); v->poseidon_sum = rd_qm31(&r);
    v->cfg.pow_bits = rd_u32(&r);
    v->cfg.log_blowup_factor = rd_u32(&r);
    v->cfg.log_last_layer_degree_bound = rd_u32(&r);
    uint64_t nq = rd_u64(&r);
    if (!r.ok || nq == 0 || nq > MAX_QUERIES) return 0;
    v->cfg.n_queries = (uint32_t)nq;
    if (want && memcmp(want, &v->cfg, sizeof *want) != 0) return 0;
    uint32_t b = v->cfg.log_blowup_factor, last = v->cfg.log_last_layer_degree_bound;
    if (v->lp < 1 || v->lq < 1 || v->lp > 28 || v->lq > 28 || b < 1 || b > 16 || last > 16 ||
        v->cfg.pow_bits > 30)
        return 0;
    v->A = v->lp + b; v->B = v->lq + b;
    v->M = umax(v->lp + 1, v->lq + 2) + b;
    if (v->M > 30) return 0;
    /* every committed column must be larger than the last FRI layer */
    if (v->A < last + b + 1 || v->B < last + b + 1) return 0;

    if (rd_u64(&r) != 4) return 0;
    for (int t = 0; t < 4; t++) v->commitments[t] = rd_words(&r, 8);
    if (rd_u64(&r) != 4) return 0;
    int k = 0;
    for (int t = 0; t < 4 && r.ok; t++) {
        if (rd_u64(&r) != TREE_COLS[t]) return 0;
        for (uint32_t c = 0; c < TREE_COLS[t]; c++) {
            uint64_t ns = rd_u64(&r);
            if (!r.ok || ns != (uint64_t)n_samples_of(t, (int)c)) return 0;
            for (uint64_t s = 0; s < ns; s++) v->samples[k++] = rd_qm31(&r);
        }
    }
    if (rd_u64(&r) != 4) return 0;
    for (int t = 0; t < 4; t++) rd_decommit(&r, &v->decommit[t]);
    if (rd_u64(&r) != 4) return 0;
    for (int t = 0; t < 4; t++) {
        v->n_queried[t] = rd_u64(&r);
        if (v->n_queried[t] > (1u << 22)) return 0;
        v->queried[t] = rd_words(&r, v->n_queried[t]);
    }
    v->pow_nonce = rd_u64(&r);
    rd_fri_layer(&r, &v->first);
    uint64_t n_inner = rd_u64(&r);
    if (!r.ok || n_inner != (uint64_t)(v->M - 1 - (last + b))) return 0;
    v->n_inner = (uint32_t)n_inner;
    for (uint32_t i = 0; i < v->n_inner; i++) rd_fri_layer(&r, &v->inner[i]);
    v->n_last = rd_u64(&r);
    if (!r.ok || v->n_last != (1ull << last)) return 0; /* components/hints/src/fiat_shamir.rs:195-198 */
    v->last_coeffs = rd_words(&r, v->n_last * 4);
    v->last_log_size = rd_u32(&r);
    if (!r.ok || r.pos != len) return 0;
    return 1;
}

/* ---------------------------------------------------- a6/a7: transcript
 * FiatShamirResults::compute (components/recursive/fiat_shamir/src/lib.rs:44-130) */
typedef struct {
    qm31 z, alpha, random_coeff, oods_t, after;
    qpoint oods;
    qm31 fri_alphas[MAX_LAYERS + 1];
    uint32_t raw_queries[MAX_QUERIES];
    m31 pow_digest[8];
    int pow_ok;
} transcript;

static uint32_t sort_dedup(uint32_t* q, uint32_t n);
static channel* g_pre_nonce; /* set by rsvo_grind_nonce around run_transcript */
static void run_transcript(const proof_view* v, transcript* t) {
    channel ch; memset(&ch, 0, sizeof ch);
    qm31 dummy;
    ch_mix_root(&ch, v->commitments[0]);
    /* statement 0: components/recursive/data_structures/src/lib.rs:52-55 */
    ch_mix_one(&ch, q_from_m(v->lp));
    ch_mix_one(&ch, q_from_m(v->lq));
    ch_mix_root(&ch, v->commitments[1]);
    ch_draw(&ch, &t->z, &t->alpha); /* LookupElementsVar::draw, data_structures/src/lib.rs:242-245 */
    ch_mix_two(&ch, v->plonk_sum, v->poseidon_sum); /* statement 1: data_structures/src/lib.rs:85-87 */
    ch_mix_root(&ch, v->commitments[2]);
    ch_draw(&ch, &t->random_coeff, &dummy);
    ch_mix_root(&ch, v->commitments[3]);
    ch_draw(&ch, &t->oods_t, &dummy);
    { /* CirclePointQM31Var::from_t, primitives/circle/src/lib.rs:204-219 */
        qm31 t2 = q_mul(t->oods_t, t->oods_t);
        qm31 inv = q_inv(q_add(t2, Q_ONE));
        t->oods.x = q_mul(q_sub(Q_ONE, t2), inv);
        t->oods.y = q_mul(q_add(t->oods_t, t->oods_t), inv);
    }
    for (int i = 0; i < N_SAMPLES; i += 2) /* fiat_shamir/src/lib.rs:68-75 */
        ch_mix_two(&ch, v->samples[i], v->samples[i + 1]);
    ch_draw(&ch, &t->after, &dummy);
    ch_mix_root(&ch, v->first.commitment);
    ch_draw(&ch, &t->fri_alphas[0], &dummy);
    for (uint32_t i = 0; i < v->n_inner; i++) {
        ch_mix_root(&ch, v->inner[i].commitment);
        ch_draw(&ch, &t->fri_alphas[i + 1], &dummy);
    }
    for (uint64_t i = 0; i < v->n_last; i += 2) { /* fiat_shamir/src/lib.rs:94-100 */
        const uint32_t* c = v->last_coeffs + 4 * i;
        if (i + 1 < v->n_last) ch_mix_two(&ch, q_mk(c[0], c[1], c[2], c[3]), q_mk(c[4], c[5], c[6], c[7]));
        else ch_mix_one(&ch, q_mk(c[0], c[1], c[2], c[3]));
    }
    if (g_pre_nonce) *g_pre_nonce = ch; /* test helper rsvo_grind_nonce: channel state before the nonce is mixed */
    /* nonce split 22/21/21 bits: data_structures/src/lib.rs:197-213, fiat_shamir/src/lib.rs:102-113 */
    uint64_t n = v->pow_nonce;
    ch_mix_one(&ch, q_mk((m31)(n & ((1u << 22) - 1)), (m31)((n >> 22) & ((1u << 21) - 1)),
                         (m31)((n >> 43) & ((1u << 21) - 1)), 0));
    memcpy(t->pow_digest, ch.digest, 32);
    t->pow_ok = (ch.digest[0] & ((1u << v->cfg.pow_bits) - 1)) == 0; /* fiat_shamir/src/lib.rs:115-117 */
    uint32_t nq = v->cfg.n_queries, got = 0; /* fiat_shamir/src/lib.rs:119-130 */
    uint32_t draws = 0;
    while (got < nq) {
        qm31 a, b2;
        ch_draw(&ch, &a, &b2);
        draws++;
        uint32_t w[8] = {a.a.a, a.a.b, a.b.a, a.b.b, b2.a.a, b2.a.b, b2.b.a, b2.b.b};
        for (int i = 0; i < 8 && got < nq; i++) t->raw_queries[got++] = w[i];
    }
    /* The circuit draws ceil(n_q / 4) times and truncates (fiat_shamir/src/lib.rs:119-130: one draw = two felts = 8
     * words, so ceil(n_q / 8) draws already hold every query): the surplus draws change no value, but each is a
     * Poseidon invocation of the circuit — replayed only when the flow is being recorded. */
    if (g_flow)
        for (; draws < (nq + 3) / 4; draws++) { qm31 a, b2; ch_draw(&ch, &a, &b2); }
}

/* Test-vector helper (not part of the verify path): find a proof-of-work nonce >= start for the proof as it stands
 * (e.g. after a test has changed a sampled value), optionally one whose query positions contain a duplicate, so
 * that rejection paths BEHIND the proof-of-work check can be exercised.  Returns RSV_OK and *nonce, or RSV_E_SIZE
 * when the proof does not parse / no nonce below start + max_tries qualifies. */
int rsvo_grind_nonce(const uint8_t* proof, size_t len, uint64_t start, uint64_t max_tries, int want_duplicate_query,
                     uint64_t* nonce) {
    if (!proof || !nonce) return RSV_E_NULL;
    proof_view* v = malloc(sizeof *v);
    transcript* t = malloc(sizeof *t);
    channel pre;
    int rc = RSV_E_SIZE;
    if (!parse_proof(proof, len, NULL, v)) goto done;
    g_pre_nonce = &pre;
    run_transcript(v, t);
    g_pre_nonce = NULL;
    const uint32_t mask = (1u << v->cfg.pow_bits) - 1, nq = v->cfg.n_queries, qmask = (1u << v->M) - 1;
    for (uint64_t n = start; n < start + max_tries; n++) {
        channel ch = pre;
        ch_mix_one(&ch, q_mk((m31)(n & ((1u << 22) - 1)), (m31)((n >> 22) & ((1u << 21) - 1)),
                             (m31)((n >> 43) & ((1u << 21) - 1)), 0));
        if (ch.digest[0] & mask) continue;
        if (want_duplicate_query) {
            uint32_t q[MAX_QUERIES], got = 0;
            while (got < nq) {
                qm31 a, b2;
                ch_draw(&ch, &a, &b2);
                uint32_t w[8] = {a.a.a, a.a.b, a.b.a, a.b.b, b2.a.a, b2.a.b, b2.b.a, b2.b.b};
                for (int i = 0; i < 8 && got < nq; i++) q[got++] = w[i] & qmask;
            }
            if (sort_dedup(q, nq) == nq) continue;
        }
        *nonce = n;
        rc = RSV_OK;
        break;
    }
done:
    free(v); free(t);
    return rc;
}

int rsvo_transcript(const uint8_t* proof, size_t len, uint32_t* out, size_t cap) {
    if (!proof || !out) return RSV_E_NULL;
    proof_view* v = malloc(sizeof *v);
    transcript* t = malloc(sizeof *t);
    int rc = RSV_OK;
    if (cap < 1) { rc = RSV_E_CAP; goto done; }
    if (!parse_proof(proof, len, NULL, v)) { out[0] = RSV_R_PARSE; goto done; }
    size_t need = 40 + 4 * (size_t)(v->n_inner + 1) + v->cfg.n_queries;
    if (cap < need) { rc = RSV_E_CAP; goto done; }
    run_transcript(v, t);
    out[0] = t->pow_ok ? RSV_R_OK : RSV_R_POW;
    out[1] = v->n_inner + 1; out[2] = v->cfg.n_queries; out[3] = v->M;
#define PUTQ(off, q) do { out[off] = (q).a.a; out[off + 1] = (q).a.b; out[off + 2] = (q).b.a; out[off + 3] = (q).b.b; } while (0)
    PUTQ(4, t->z); PUTQ(8, t->alpha); PUTQ(12, t->random_coeff); PUTQ(16, t->oods_t);
    PUTQ(20, t->oods.x); PUTQ(24, t->oods.y); PUTQ(28, t->after);
    memcpy(out + 32, t->pow_digest, 32);
    for (uint32_t i = 0; i <= v->n_inner; i++) PUTQ(40 + 4 * i, t->fri_alphas[i]);
    memcpy(out + 40 + 4 * (v->n_inner + 1), t->raw_queries, 4 * v->cfg.n_queries);
done:
    free(v); free(t);
    return rc;
}

/* ------------------------------------------------- a10: OODS composition
 * components/recursive/composition/src/{lib.rs,data_structures.rs,plonk.rs,poseidon.rs} */
typedef struct {
    qm31 rc, acc, dinv, z, alpha, alpha2, shift;
    qm31 frac_p[8], frac_q[8]; int n_fracs;
    const qm31* inter; /* this component's interaction samples, in order */
    int inter_pos;
} eval_ctx;
/* data_structures.rs:26-28,166-169 */
static void add_constraint(eval_ctx* e, qm31 c) { e->acc = q_add(q_mul(e->acc, e->rc), q_mul(c, e->dinv)); }
/* data_structures.rs:147-164 */
static void add_relation(eval_ctx* e, qm31 mult, const qm31* vals, int n) {
    qm31 d = vals[0];
    if (n > 1) d = q_add(d, q_mul(e->alpha, vals[1]));
    if (n > 2) d = q_add(d, q_mul(e->alpha2, vals[2]));
    e->frac_p[e->n_fracs] = mult;
    e->frac_q[e->n_fracs++] = q_sub(d, e->z);
}
/* data_structures.rs:171-210.  The interaction samples of one component are
 * 4 single-sample columns per non-final batch followed by 4 two-sample
 * ([-1, 0]) columns for the final batch. */
static void finalize_logup(eval_ctx* e, int batch) {
    int n_batches = (e->n_fracs + batch - 1) / batch;
    qm31 prev = Q_ZERO;
    for (int bi = 0; bi < n_batches; bi++) {
        int lo = bi * batch, hi = lo + batch < e->n_fracs ? lo + batch : e->n_fracs;
        qm31 p = e->frac_p[lo], q = e->frac_q[lo];
        for (int k = lo + 1; k < hi; k++) {
            p = q_add(q_mul(p, e->frac_q[k]), q_mul(e->frac_p[k], q));
            q = q_mul(q, e->frac_q[k]);
        }
        const qm31* s = e->inter + e->inter_pos;
        if (bi < n_batches - 1) {
            qm31 cur = q_combine_ef(s[0], s[1], s[2], s[3]);
            e->inter_pos += 4;
            add_constraint(e, q_sub(q_mul(q_sub(cur, prev), q), p));
            prev = cur;
        } else {
            qm31 prev_row = q_combine_ef(s[0], s[2], s[4], s[6]);
            qm31 cur = q_combine_ef(s[1], s[3], s[5], s[7]);
            e->inter_pos += 8;
            qm31 diff = q_sub(q_sub(cur, prev_row), prev);
            add_constraint(e, q_sub(q_mul(q_add(diff, e->shift), q), p));
        }
    }
}
/* plonk.rs:8-82 */
static void evaluate_plonk(eval_ctx* e, const qm31* pre, const qm31* tr) {
    qm31 a_wire = pre[0], b_wire = pre[1], c_wire = pre[2], op = pre[3], mult_a = pre[4], mult_b = pre[5],
         mult_c = pre[6], poseidon_wire = pre[7], mult_poseidon = pre[8], enforce_c_m31 = pre[9];
    add_constraint(e, q_mul(enforce_c_m31, tr[9]));
    add_constraint(e, q_mul(enforce_c_m31, tr[10]));
    add_constraint(e, q_mul(enforce_c_m31, tr[11]));
    qm31 a = q_combine_ef(tr[0], tr[1], tr[2], tr[3]);
    qm31 b = q_combine_ef(tr[4], tr[5], tr[6], tr[7]);
    qm31 c = q_combine_ef(tr[8], tr[9], tr[10], tr[11]);
    add_constraint(e, q_sub(q_sub(c, q_mul(op, q_add(a, b))), q_mul(q_mul(q_sub(Q_ONE, op), a), b)));
    qm31 v[3];
    e->n_fracs = 0;
    v[0] = a; v[1] = a_wire; add_relation(e, mult_a, v, 2);
    v[0] = b; v[1] = b_wire; add_relation(e, mult_b, v, 2);
    v[0] = c; v[1] = c_wire; add_relation(e, mult_c, v, 2);
    v[0] = poseidon_wire; v[1] = a; v[2] = b; add_relation(e, q_neg(mult_poseidon), v, 3);
    finalize_logup(e, 2);
}
/* poseidon.rs:12-71: the same round functions over QM31 */
static void q_m4(qm31* x) {
    qm31 t0 = q_add(x[0], x[1]), t02 = q_add(t0, t0), t1 = q_add(x[2], x[3]), t12 = q_add(t1, t1);
    qm31 t2 = q_add(q_add(x[1], x[1]), t1), t3 = q_add(q_add(x[3], x[3]), t0);
    qm31 t4 = q_add(q_add(t12, t12), t3), t5 = q_add(q_add(t02, t02), t2);
    x[0] = q_add(t3, t5); x[1] = t5; x[2] = q_add(t2, t4); x[3] = t4;
}
static void q_external(qm31* s) {
    for (int g = 0; g < 4; g++) q_m4(s + 4 * g);
    for (int j = 0; j < 4; j++) {
        qm31 sum = q_add(q_add(s[j], s[j + 4]), q_add(s[j + 8], s[j + 12]));
        for (int g = 0; g < 4; g++) s[4 * g + j] = q_add(s[4 * g + j], sum);
    }
}
static void q_internal(qm31* s) {
    qm31 sum = s[0];
    for (int i = 1; i < 16; i++) sum = q_add(sum, s[i]);
    s[0] = q_add(s[0], q_add(q_add(s[0], s[0]), sum));
    for (int i = 1; i < 16; i++) s[i] = q_add(q_mul_m(s[i], 1u << (i + 1)), sum);
}
static qm31 q_pow5(qm31 x) { qm31 x2 = q_mul(x, x); return q_mul(q_mul(x2, x2), x); }
/* poseidon.rs:73-241 */
static void evaluate_poseidon(eval_ctx* e, const qm31* pre, const qm31* tr) {
    qm31 is_first = pre[0], is_last = pre[1], is_full = pre[2], round_id = pre[3];
    const qm31 *rc0 = pre + 4, *rc1 = pre + 20;
    qm31 ext1 = pre[36], ext2 = pre[37], ext1_nz = pre[38], ext2_nz = pre[39];
    qm31 not_first = q_sub(Q_ONE, is_first), not_last = q_sub(Q_ONE, is_last);
    qm31 is_partial = q_sub(not_first, is_full);
    const qm31 *in = tr, *mid = tr + 16, *out = tr + 32;
    qm31 swap_addr = rc0[0], swap_val = mid[0], one_minus_swap = q_sub(Q_ONE, swap_val);
    qm31 st[16];
    for (int i = 0; i < 16; i++)
        st[i] = i < 8 ? q_add(q_mul(in[i], one_minus_swap), q_mul(in[i + 8], swap_val))
                      : q_add(q_mul(in[i - 8], swap_val), q_mul(in[i], one_minus_swap));
    q_external(st);
    for (int i = 0; i < 16; i++) add_constraint(e, q_mul(is_first, q_sub(st[i], out[i])));
    /* full round */
    for (int i = 0; i < 16; i++) st[i] = q_pow5(q_add(in[i], rc0[i]));
    for (int i = 0; i < 16; i++) {
        add_constraint(e, q_mul(is_full, q_sub(mid[i], st[i])));
        st[i] = mid[i];
    }
    q_external(st);
    for (int i = 0; i < 16; i++) st[i] = q_pow5(q_add(st[i], rc1[i]));
    q_external(st);
    for (int i = 0; i < 16; i++) add_constraint(e, q_mul(is_full, q_sub(out[i], st[i])));
    /* partial rounds */
    for (int i = 0; i < 16; i++) st[i] = in[i];
    for (int r = 0; r < 14; r++) {
        st[0] = q_pow5(q_add(st[0], rc0[r]));
        add_constraint(e, q_mul(is_partial, q_sub(mid[r], st[0])));
        st[0] = mid[r];
        q_internal(st);
    }
    for (int i = 0; i < 16; i++) add_constraint(e, q_mul(is_partial, q_sub(out[i], st[i])));
    /* lookups */
    qm31 in_left = q_add(round_id, round_id), in_right = q_add(in_left, Q_ONE);
    qm31 out_left = q_add(in_right, Q_ONE), out_right = q_add(out_left, Q_ONE);
    qm31 v[3];
    e->n_fracs = 0;
    v[0] = q_add(q_mul(is_first, ext1), q_mul(not_first, in_left));
    v[1] = q_combine_ef(in[0], in[1], in[2], in[3]); v[2] = q_combine_ef(in[4], in[5], in[6], in[7]);
    add_relation(e, q_sub(q_mul(ext1_nz, is_first), not_first), v, 3);
    v[0] = q_add(q_mul(is_first, ext2), q_mul(not_first, in_right));
    v[1] = q_combine_ef(in[8], in[9], in[10], in[11]); v[2] = q_combine_ef(in[12], in[13], in[14], in[15]);
    add_relation(e, q_sub(q_mul(ext2_nz, is_first), not_first), v, 3);
    v[0] = q_add(q_mul(is_last, ext1), q_mul(not_last, out_left));
    v[1] = q_combine_ef(out[0], out[1], out[2], out[3]); v[2] = q_combine_ef(out[4], out[5], out[6], out[7]);
    add_relation(e, q_add(q_mul(ext1_nz, is_last), not_last), v, 3);
    v[0] = q_add(q_mul(is_last, ext2), q_mul(not_last, out_right));
    v[1] = q_combine_ef(out[8], out[9], out[10], out[11]); v[2] = q_combine_ef(out[12], out[13], out[14], out[15]);
    add_relation(e, q_add(q_mul(ext2_nz, is_last), not_last), v, 3);
    v[0] = swap_val; v[1] = swap_addr;
    add_relation(e, q_mul(is_first, not_last), v, 2);
    finalize_logup(e, 3);
}

/* sample offsets of each tree inside proof_view.samples */
enum { S_T0 = 0, S_T1 = 50, S_T2 = 110, S_T3 = 134 };

/* CompositionCheck::compute (composition/src/lib.rs:34-121): the accumulator over the 86 constraints and the value
 * the committed composition polynomial gives at the OODS point */
static void eval_composition(const proof_view* v, const transcript* t, qm31* acc_out, qm31* expected_out) {
    eval_ctx e; memset(&e, 0, sizeof e);
    e.rc = t->random_coeff; e.z = t->z; e.alpha = t->alpha; e.alpha2 = q_mul(t->alpha, t->alpha);
    /* plonk: pre cols 0..10, trace 0..12, interaction samples 0..12 (4 + 4*2) */
    /* coset_vanishing (composition/src/lib.rs:18-29): Z_H(p) = pi^(n-1)(p.x) */
    e.dinv = q_inv(q_double_x(t->oods.x, v->lp - 1));
    e.shift = q_mul_m(v->plonk_sum, m_inv(1u << v->lp)); /* data_structures.rs:67-68 */
    e.inter = v->samples + S_T2; e.inter_pos = 0;
    evaluate_plonk(&e, v->samples + S_T0, v->samples + S_T1);
    e.dinv = q_inv(q_double_x(t->oods.x, v->lq - 1));
    e.shift = q_mul_m(v->poseidon_sum, m_inv(1u << v->lq));
    e.inter = v->samples + S_T2 + 12; e.inter_pos = 0;
    evaluate_poseidon(&e, v->samples + S_T0 + 10, v->samples + S_T1 + 12);
    const qm31* c = v->samples + S_T3;
    qm31 left = q_combine_ef(c[0], c[1], c[2], c[3]), right = q_combine_ef(c[4], c[5], c[6], c[7]);
    uint32_t comp_log_degree_bound = umax(v->lp + 2, v->lq + 3);
    *expected_out = q_add(left, q_mul(right, q_double_x(t->oods.x, comp_log_degree_bound - 2)));
    *acc_out = e.acc;
}
static int check_composition(const proof_view* v, const transcript* t) {
    qm31 acc, expected;
    eval_composition(v, t, &acc, &expected);
    return q_eq(acc, expected);
}
/* Probe (rsv_oods_eval's counterpart): item i = 142 sampled values (4 words each) + 26 parameter words
 * lp, lq, plonk_sum, poseidon_sum, z, alpha, random_coeff, oods.x; out = accumulator | expected (8 words). */
int rsvo_oods_eval(const uint32_t* samples4, const uint32_t* params26, uint32_t* out8, size_t n) {
    if (n && (!samples4 || !params26 || !out8)) return RSV_E_NULL;
    if (!canonical(samples4, n * N_SAMPLES * 4)) return RSV_E_RANGE;
    proof_view* v = calloc(1, sizeof *v);
    transcript* t = calloc(1, sizeof *t);
    int rc = RSV_OK;
    for (size_t i = 0; i < n && rc == RSV_OK; i++) {
        const uint32_t* pr = params26 + 26 * i;
        if (pr[0] < 1 || pr[0] > 28 || pr[1] < 1 || pr[1] > 28 || !canonical(pr + 2, 24)) { rc = RSV_E_RANGE; break; }
        v->lp = pr[0]; v->lq = pr[1];
        v->plonk_sum = q_mk(pr[2], pr[3], pr[4], pr[5]); v->poseidon_sum = q_mk(pr[6], pr[7], pr[8], pr[9]);
        t->z = q_mk(pr[10], pr[11], pr[12], pr[13]); t->alpha = q_mk(pr[14], pr[15], pr[16], pr[17]);
        t->random_coeff = q_mk(pr[18], pr[19], pr[20], pr[21]); t->oods.x = q_mk(pr[22], pr[23], pr[24], pr[25]);
        for (int k = 0; k < N_SAMPLES; k++) {
            const uint32_t* w = samples4 + (i * N_SAMPLES + (size_t)k) * 4;
            v->samples[k] = q_mk(w[0], w[1], w[2], w[3]);
        }
        qm31 acc, expected;
        eval_composition(v, t, &acc, &expected);
        uint32_t* o = out8 + 8 * i;
        o[0] = acc.a.a; o[1] = acc.a.b; o[2] = acc.b.a; o[3] = acc.b.b;
        o[4] = expected.a.a; o[5] = expected.a.b; o[6] = expected.b.a; o[7] = expected.b.b;
    }
    free(v); free(t);
    return rc;
}

/* logup total-sum check, fiat_shamir/src/lib.rs:133-141 */
static int check_logup(const proof_view* v, const transcript* t, const rsv_public_input* pi, size_t n_pi) {
    qm31 sum = Q_ZERO;
    for (size_t i = 0; i < n_pi; i++) {
        qm31 val = q_mk(pi[i].value[0], pi[i].value[1], pi[i].value[2], pi[i].value[3]);
        qm31 d = q_sub(q_add(val, q_mul(q_from_m(pi[i].idx % P), t->alpha)), t->z);
        sum = q_add(sum, q_inv(d));
    }
    return q_eq(q_add(q_add(sum, v->poseidon_sum), v->plonk_sum), Q_ZERO);
}

/* ----------------------------------------------------------- verification */
typedef struct { uint32_t pos; m31 h[8]; } node;

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : x > y;
}
static uint32_t sort_dedup(uint32_t* q, uint32_t n) {
    qsort(q, n, 4, cmp_u32);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; i++) if (i == 0 || q[i] != q[k - 1]) q[k++] = q[i];
    return k;
}
static const node* find_node(const node* l, uint32_t n, uint32_t pos) {
    for (uint32_t i = 0; i < n; i++) if (l[i].pos == pos) return &l[i];
    return NULL;
}

/* Batched decommitment of one trace tree: the walk of
 * SinglePathMerkleProof::from_stwo_proof (components/hints/src/decommit.rs:53-142).
 * ncols_at[h] = number of columns committed at log size h in this tree.
 * cols_out[h] (if non-NULL) receives, for the distinct positions at layer h in
 * ascending order, a pointer to that node's column words. */
typedef struct { uint32_t n; uint32_t pos[MAX_QUERIES]; const uint32_t* vals[MAX_QUERIES]; } layer_cols;
/* every node hash known at a layer (computed nodes and witness siblings), for per-path extraction */
typedef struct { uint32_t n[32]; node nodes[32][2 * MAX_QUERIES]; } tree_record;
static void rec_add(tree_record* r, uint32_t layer, uint32_t pos, const m31* h) {
    if (!r) return;
    node* d = &r->nodes[layer][r->n[layer]++];
    d->pos = pos; memcpy(d->h, h, 32);
}
static tree_record* g_record; /* set by rsvo_trace_paths around verify_trace_tree */

static int verify_trace_tree(const uint32_t* queries_at_max, uint32_t nq, uint32_t maxlog,
                             const uint32_t* ncols_at, const uint32_t* values, uint64_t n_values,
                             const decommit_view* dec, const uint32_t* root, layer_cols* cols_out) {
    uint32_t pos[MAX_QUERIES];
    memcpy(pos, queries_at_max, 4 * nq);
    uint32_t n = sort_dedup(pos, nq);
    uint64_t vi = 0, hi = 0;
    node cur[2 * MAX_QUERIES], nxt[MAX_QUERIES];
    uint32_t ncur = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t nc = ncols_at[maxlog];
        if (vi + nc > n_values) return 0;
        cur[ncur].pos = pos[i];
        hash_node(NULL, NULL, values + vi, nc, cur[ncur].h);
        rec_add(g_record, maxlog, pos[i], cur[ncur].h);
        if (cols_out) { cols_out[maxlog].pos[i] = pos[i]; cols_out[maxlog].vals[i] = values + vi; }
        vi += nc; ncur++;
    }
    if (cols_out) cols_out[maxlog].n = n;
    for (uint32_t layer = maxlog; layer-- > 0;) {
        uint32_t nn = 0, nc = ncols_at[layer];
        uint32_t nknown = ncur; /* siblings from the witness are appended after the known nodes */
        for (uint32_t i = 0; i < n; i++) {
            uint32_t parent = pos[i] >> 1;
            if (nn && nxt[nn - 1].pos == parent) continue;
            if (vi + nc > n_values) return 0;
            const uint32_t* c = values + vi; vi += nc;
            const node* self = find_node(cur, nknown, pos[i]);
            const node* sib = find_node(cur, nknown, pos[i] ^ 1);
            const m31* sh;
            if (sib) sh = sib->h;
            else {
                if (hi >= dec->n_hash) return 0;
                sh = dec->hash_witness + 8 * hi++;
                rec_add(g_record, layer + 1, pos[i] ^ 1, sh);
            }
            nxt[nn].pos = parent;
            if (pos[i] & 1) hash_node(sh, self->h, c, nc, nxt[nn].h);
            else hash_node(self->h, sh, c, nc, nxt[nn].h);
            rec_add(g_record, layer, parent, nxt[nn].h);
            if (cols_out && nc) { cols_out[layer].pos[nn] = parent; cols_out[layer].vals[nn] = c; }
            nn++;
        }
        if (cols_out && nc) cols_out[layer].n = nn;
        for (uint32_t i = 0; i < nn; i++) { cur[i] = nxt[i]; pos[i] = nxt[i].pos; }
        n = ncur = nn;
    }
    if (hi != dec->n_hash || vi != n_values) return 0; /* decommit.rs:141-142 */
    return n == 1 && memcmp(cur[0].h, root, 32) == 0;
}

/* Batched decommitment of a FRI "pair" tree: SinglePairMerkleProof::from_stwo_proof
 * (components/hints/src/folding.rs:93-212).  has_data[l] marks layers carrying one
 * QM31 column; values lists, per data layer (descending), for the sorted
 * positions (queries and their siblings) 4 words each. */
/* optional recording for rsvo_fri_paths: every node hash known at a level, the column value of every data-level
 * candidate, and for data levels below the leaves the hash of the candidate's children (hash_node(children, [])) */
typedef struct {
    uint32_t n[32], nc[32];
    node nodes[32][8 * MAX_QUERIES];
    struct { uint32_t pos; m31 val[4]; m31 kids[8]; } cand[32][2 * MAX_QUERIES];
} pair_record;
static pair_record* g_pair_record;

static int verify_pair_tree(const uint32_t* leaf_queries, uint32_t nq, uint32_t maxlog,
                            const uint8_t* has_data, const uint32_t* values, uint64_t n_values,
                            const decommit_view* dec, const uint32_t* root) {
    pair_record* rec = g_pair_record;
    if (rec) { memset(rec->n, 0, sizeof rec->n); memset(rec->nc, 0, sizeof rec->nc); }
    uint32_t q[MAX_QUERIES];
    memcpy(q, leaf_queries, 4 * nq);
    uint32_t n = nq;
    uint64_t vi = 0, hi = 0;
    node prev[8 * MAX_QUERIES], cur[2 * MAX_QUERIES];
    uint32_t nprev = 0;
    for (uint32_t l = maxlog + 1; l-- > 0;) {
        n = sort_dedup(q, n);
        uint32_t ncur = 0;
        uint32_t cand[2 * MAX_QUERIES], ncand = 0;
        if (has_data[l]) {
            for (uint32_t i = 0; i < n; i++) { cand[ncand++] = q[i]; cand[ncand++] = q[i] ^ 1; }
            ncand = sort_dedup(cand, ncand);
        } else {
            if (l == maxlog) return 0;
            memcpy(cand, q, 4 * n); ncand = n;
        }
        for (uint32_t i = 0; i < ncand; i++) {
            const uint32_t* val = NULL;
            if (has_data[l]) {
                if (vi + 4 > n_values) return 0;
                val = values + vi; vi += 4;
            }
            cur[ncur].pos = cand[i];
            if (l == maxlog) hash_node(NULL, NULL, val, 4, cur[ncur].h);
            else {
                const m31 *lh, *rh;
                const node* ln = find_node(prev, nprev, cand[i] << 1);
                if (ln) lh = ln->h;
                else {
                    if (hi >= dec->n_hash) return 0;
                    lh = dec->hash_witness + 8 * hi++;
                    prev[nprev].pos = cand[i] << 1; memcpy(prev[nprev].h, lh, 32); nprev++;
                }
                const node* rn = find_node(prev, nprev, (cand[i] << 1) + 1);
                if (rn) rh = rn->h;
                else {
                    if (hi >= dec->n_hash) return 0;
                    rh = dec->hash_witness + 8 * hi++;
                    prev[nprev].pos = (cand[i] << 1) + 1; memcpy(prev[nprev].h, rh, 32); nprev++;
                }
                hash_node(lh, rh, val, val ? 4 : 0, cur[ncur].h);
                if (rec && val) hash_node(lh, rh, NULL, 0, rec->cand[l][rec->nc[l]].kids);
            }
            if (rec && val) {
                rec->cand[l][rec->nc[l]].pos = cand[i];
                memcpy(rec->cand[l][rec->nc[l]].val, val, 16);
                rec->nc[l]++;
            }
            ncur++;
        }
        if (rec && l < maxlog) { /* children level l+1 is complete now (witness children were appended to prev) */
            memcpy(rec->nodes[l + 1], prev, sizeof(node) * nprev); rec->n[l + 1] = nprev;
        }
        memcpy(prev, cur, sizeof(node) * ncur); nprev = ncur;
        for (uint32_t i = 0; i < n; i++) q[i] >>= 1;
    }
    if (vi != n_values || hi != dec->n_hash) return 0; /* folding.rs:208-209 */
    return nprev == 1 && memcmp(prev[0].h, root, 32) == 0; /* folding.rs:211-212 */
}

/* DEEP quotients for one column-log-size group
 * (components/recursive/answer/src/data_structures.rs:43-189, src/lib.rs:356-382). */
typedef struct { int col; qm31 value; } col_sample;
typedef struct { qpoint point; int n; col_sample cs[N_SAMPLES]; qm31 a[N_SAMPLES], b[N_SAMPLES], c[N_SAMPLES]; } sample_batch;

typedef struct {
    uint32_t log_size;
    int n_cols;
    int n_batches;
    sample_batch batch[2];
} quotient_group;

/* Adds the samples of tree `t` columns [c0, c1) to the group; shift_log != 0
 * only matters for the two-sample interaction columns. */
static void group_add_cols(quotient_group* g, const proof_view* v, const qpoint* oods, const qpoint* shifted,
                           int t, int c0, int c1) {
    static const int tree_sample_base[4] = {S_T0, S_T1, S_T2, S_T3};
    for (int c = c0; c < c1; c++) {
        /* flattened sample index of (t, c) */
        int si = tree_sample_base[t] + c;
        if (t == 2) si = S_T2 + (c < 4 ? c : c < 8 ? 4 + 2 * (c - 4) : c < 12 ? 12 + (c - 8) : 16 + 2 * (c - 12));
        int col = g->n_cols++;
        int ns = n_samples_of(t, c);
        for (int s = 0; s < ns; s++) {
            int shifted_sample = (ns == 2 && s == 0); /* mask offsets [-1, 0] */
            /* IndexMap insertion order: the Zero batch is created by the first
             * (preprocessed / composition) column, the shift batch afterwards. */
            int bi = shifted_sample ? 1 : 0;
            if (bi == 1 && g->n_batches < 2) { g->n_batches = 2; g->batch[1].point = *shifted; g->batch[1].n = 0; }
            if (bi == 0 && g->n_batches < 1) { g->n_batches = 1; g->batch[0].point = *oods; g->batch[0].n = 0; }
            sample_batch* b = &g->batch[bi];
            b->cs[b->n].col = col; b->cs[b->n].value = v->samples[si + s]; b->n++;
        }
    }
}
/* column_line_coeffs_var / complex_conjugate_line_coeffs_var (data_structures.rs:132-189) */
static void group_line_coeffs(quotient_group* g, qm31 after) {
    qm31 alpha = q_mk(0, 0, m_neg(2), 0);
    for (int bi = 0; bi < g->n_batches; bi++) {
        sample_batch* b = &g->batch[bi];
        cm31 y0 = b->point.y.a, y1 = b->point.y.b;
        for (int k = 0; k < b->n; k++) {
            cm31 v0 = b->cs[k].value.a, v1 = b->cs[k].value.b;
            b->a[k] = q_mul_c(alpha, v1);
            b->b[k] = q_mul_c(alpha, c_sub(c_mul(v0, y1), c_mul(v1, y0)));
            b->c[k] = q_mul_c(alpha, y1);
            alpha = q_mul(alpha, after);
        }
    }
}
/* accumulate_row_quotients_var + denominator_inverses_var (data_structures.rs:70-130) */
static qm31 row_quotient(const quotient_group* g, const m31* row, cpoint dom) {
    qm31 acc = Q_ZERO;
    for (int bi = 0; bi < g->n_batches; bi++) {
        const sample_batch* b = &g->batch[bi];
        qm31 num = Q_ZERO;
        for (int k = 0; k < b->n; k++) {
            qm31 value = q_mul_m(b->c[k], row[b->cs[k].col]);
            qm31 linear = q_add(q_mul_m(b->a[k], dom.y), b->b[k]);
            num = q_add(num, q_sub(value, linear));
        }
        cm31 prx = b->point.x.a, pix = b->point.x.b, pry = b->point.y.a, piy = b->point.y.b;
        cm31 den = c_sub(c_mul(c_sub(prx, c_mk(dom.x, 0)), piy), c_mul(c_sub(pry, c_mk(dom.y, 0)), pix));
        acc = q_add(acc, q_mul_c(num, c_inv(den)));
    }
    return acc;
}

static const uint32_t* find_cols(const layer_cols* lc, uint32_t pos) {
    for (uint32_t i = 0; i < lc->n; i++) if (lc->pos[i] == pos) return lc->vals[i];
    return NULL;
}

/* LinePolyVar::eval_at_point (primitives/line/src/lib.rs:39-67) */
static qm31 line_fold(const uint32_t* coeffs, uint64_t n, const m31* factors) {
    if (n == 1) return q_mk(coeffs[0], coeffs[1], coeffs[2], coeffs[3]);
    qm31 l = line_fold(coeffs, n / 2, factors + 1), r = line_fold(coeffs + 4 * (n / 2), n / 2, factors + 1);
    return q_add(l, q_mul_m(r, factors[0]));
}

/* probes with the op codes of include/rsv.h (RSV_F_*) */
int rsvo_field_op(int op, const uint32_t* a4, const uint32_t* b4, uint32_t* out4, size_t n) {
    if (n && (!a4 || !out4)) return RSV_E_NULL;
    for (size_t i = 0; i < n; i++) {
        qm31 x = q_mk(a4[4 * i], a4[4 * i + 1], a4[4 * i + 2], a4[4 * i + 3]);
        qm31 y = b4 ? q_mk(b4[4 * i], b4[4 * i + 1], b4[4 * i + 2], b4[4 * i + 3]) : Q_ZERO, r = Q_ZERO;
        switch (op) {
            case 0: r = q_add(x, y); break;
            case 1: r = q_sub(x, y); break;
            case 2: r = q_mul(x, y); break;
            case 3: r = q_inv(x); break;
            case 4: r = q_mk(m_mul(x.a.a, y.a.a), 0, 0, 0); break;
            case 5: r = q_mk(m_inv(x.a.a), 0, 0, 0); break;
            case 6: { cm31 c = c_mul(x.a, y.a); r = q_mk(c.a, c.b, 0, 0); } break;
            case 7: { cm31 c = c_inv(x.a); r = q_mk(c.a, c.b, 0, 0); } break;
            case 8: { const qm31 I = {{0, 1}, {0, 0}}; r = q_mul(x, I); } break;
            case 9: { const qm31 U = {{0, 0}, {1, 0}}; r = q_mul(x, U); } break;
            case 10: { /* square-and-multiply from the top bit, unlike the GPU's bottom-up loop */
                r = q_mk(1, 0, 0, 0);
                for (int bit = 31; bit >= 0; bit--) { r = q_mul(r, r); if ((y.a.a >> bit) & 1u) r = q_mul(r, x); }
            } break;
            default: return RSV_E_SIZE;
        }
        out4[4 * i] = r.a.a; out4[4 * i + 1] = r.a.b; out4[4 * i + 2] = r.b.a; out4[4 * i + 3] = r.b.b;
    }
    return RSV_OK;
}
int rsvo_line_eval(const uint32_t* coeffs4, uint32_t log_n, const uint32_t* x, uint32_t* out4, size_t n) {
    if (!coeffs4 || (n && (!x || !out4)) || log_n > 16) return RSV_E_NULL;
    for (size_t i = 0; i < n; i++) {
        m31 d[32], v = x[i];
        for (uint32_t k = 0; k < log_n; k++) { d[k] = v; v = m_sub(m_add(m_mul(v, v), m_mul(v, v)), 1); }
        qm31 r = line_fold(coeffs4, (uint64_t)1 << log_n, d);
        out4[4 * i] = r.a.a; out4[4 * i + 1] = r.a.b; out4[4 * i + 2] = r.b.a; out4[4 * i + 3] = r.b.b;
    }
    return RSV_OK;
}

typedef struct {
    /* per query j (transcript order) */
    qm31 answers[3][MAX_QUERIES];
    qm31 last_value[MAX_QUERIES];
    qm31 first_folded[3][MAX_QUERIES]; /* FirstLayerHints::folded_evals_by_column, per size group and query */
    qm31 inner_in[MAX_LAYERS][MAX_QUERIES]; /* value entering inner layer i (the query's own leaf of that layer's tree) */
    qm31 last_eval[MAX_QUERIES];            /* last-layer polynomial at the query's point */
    uint32_t n_sizes;
    /* optional: per-path extraction of the trace trees (rsvo_trace_paths) */
    tree_record* records; /* [4] or NULL */
    uint32_t qM[MAX_QUERIES], M, maxlog[4], nq;
    pair_record* pair_records; /* [1 + n_inner] or NULL */
    uint32_t n_inner;
    uint32_t* trace_cols; /* [4][nq][64] or NULL: per query, the leaf-level columns then the lower-level columns */
} query_probe;

static uint8_t verify_one(const uint8_t* bytes, size_t len, const rsv_pcs_config* cfg,
                          const rsv_public_input* pi, size_t n_pi, query_probe* probe) {
    uint8_t reason = RSV_R_OK;
    proof_view* v = malloc(sizeof *v);
    transcript* t = malloc(sizeof *t);
    layer_cols(*cols)[32] = calloc(4, sizeof(layer_cols[32]));
    quotient_group* groups = calloc(3, sizeof *groups);
#define FAIL(r) do { reason = (r); goto done; } while (0)
    if (!parse_proof(bytes, len, cfg, v)) FAIL(RSV_R_PARSE);
    run_transcript(v, t);
    if (!t->pow_ok) FAIL(RSV_R_POW);
    if (!check_logup(v, t, pi, n_pi)) FAIL(RSV_R_LOGUP);
    if (!check_composition(v, t)) FAIL(RSV_R_COMPOSITION);

    const uint32_t nq = v->cfg.n_queries, M = v->M, A = v->A, B = v->B;
    /* query positions: primitives/query/src/lib.rs:19-38 */
    uint32_t qM[MAX_QUERIES];
    for (uint32_t j = 0; j < nq; j++) qM[j] = t->raw_queries[j] & ((1u << M) - 1);
    { /* components/recursive/answer/src/lib.rs:190-195 */
        uint32_t tmp[MAX_QUERIES];
        memcpy(tmp, qM, 4 * nq);
        if (sort_dedup(tmp, nq) != nq) FAIL(RSV_R_DUP_QUERY);
    }
    /* trace trees: components/recursive/answer/src/lib.rs:214-258 */
    for (int tr = 0; tr < 4; tr++) {
        static const uint32_t plonk_cols[3] = {10, 12, 8}, poseidon_cols[3] = {40, 48, 8};
        uint32_t ncols_at[32] = {0}, maxlog;
        if (tr < 3) { ncols_at[A] += plonk_cols[tr]; ncols_at[B] += poseidon_cols[tr]; maxlog = umax(A, B); }
        else { ncols_at[M] = 8; maxlog = M; }
        uint32_t q[MAX_QUERIES];
        for (uint32_t j = 0; j < nq; j++) q[j] = qM[j] >> (M - maxlog);
        if (probe->records) {
            g_record = &probe->records[tr];
            memset(g_record->n, 0, sizeof g_record->n);
            probe->maxlog[tr] = maxlog; probe->M = M; probe->nq = nq;
            memcpy(probe->qM, qM, 4 * nq);
        }
        int tree_ok = verify_trace_tree(q, nq, maxlog, ncols_at, v->queried[tr], v->n_queried[tr], &v->decommit[tr],
                                        v->commitments[tr], cols[tr]);
        g_record = NULL;
        if (!tree_ok) FAIL(RSV_R_MERKLE_T0 + tr);
        if (probe->trace_cols) { /* SinglePathMerkleProof::columns (components/hints/src/decommit.rs:158-170) */
            probe->M = M; probe->nq = nq;
            for (uint32_t j = 0; j < nq; j++) {
                uint32_t* dst = probe->trace_cols + ((size_t)tr * nq + j) * 64, k = 0;
                for (uint32_t l = maxlog + 1; l-- > 0;) {
                    if (!ncols_at[l]) continue;
                    const uint32_t* src = find_cols(&cols[tr][l], q[j] >> (maxlog - l));
                    for (uint32_t e = 0; e < ncols_at[l]; e++) dst[k++] = src ? src[e] : 0xFFFFFFFFu;
                }
            }
        }
    }
    /* quotient groups by descending column log size: answer/src/lib.rs:294-315 */
    uint32_t sizes[3]; uint32_t n_sizes = 0;
    sizes[n_sizes++] = M;
    if (A == B) sizes[n_sizes++] = A;
    else { sizes[n_sizes++] = umax(A, B); sizes[n_sizes++] = A < B ? A : B; }
    cpoint step_p = cp_subgroup_gen(v->lp), step_q = cp_subgroup_gen(v->lq);
    step_p.y = m_neg(step_p.y); step_q.y = m_neg(step_q.y); /* mul_signed(-1): answer/src/lib.rs:62-72 */
    qpoint sh_p = qp_add_m(t->oods, step_p), sh_q = qp_add_m(t->oods, step_q);
    for (uint32_t gi = 0; gi < n_sizes; gi++) {
        quotient_group* g = &groups[gi];
        g->log_size = sizes[gi];
        if (sizes[gi] == M) group_add_cols(g, v, &t->oods, &t->oods, 3, 0, 8);
        else {
            int want_p = sizes[gi] == A, want_q = sizes[gi] == B;
            static const int split[3] = {10, 12, 8};
            for (int tr = 0; tr < 3; tr++) {
                if (want_p) group_add_cols(g, v, &t->oods, &sh_p, tr, 0, split[tr]);
                if (want_q) group_add_cols(g, v, &t->oods, &sh_q, tr, split[tr], (int)TREE_COLS[tr]);
            }
        }
        group_line_coeffs(g, t->after);
    }
    /* answers per query: answer/src/lib.rs:260-315,356-382 */
    qm31(*answers)[MAX_QUERIES] = probe->answers;
    probe->n_sizes = n_sizes;
    for (uint32_t gi = 0; gi < n_sizes; gi++) {
        uint32_t l = sizes[gi];
        for (uint32_t j = 0; j < nq; j++) {
            uint32_t ql = qM[j] >> (M - l);
            m31 row[N_COLS_TOTAL]; int n = 0;
            for (int tr = 0; tr < 4; tr++) {
                uint32_t maxlog = tr < 3 ? umax(A, B) : M;
                if (l > maxlog) continue;
                uint32_t nc = 0;
                static const uint32_t plonk_cols[3] = {10, 12, 8}, poseidon_cols[3] = {40, 48, 8};
                if (tr < 3) nc = (l == A ? plonk_cols[tr] : 0) + (l == B ? poseidon_cols[tr] : 0);
                else nc = l == M ? 8 : 0;
                if (!nc) continue;
                const uint32_t* c = find_cols(&cols[tr][l], ql);
                memcpy(row + n, c, 4 * nc); n += (int)nc;
            }
            answers[gi][j] = row_quotient(&groups[gi], row, domain_point(l, ql));
        }
    }
    /* FRI first layer: components/hints/src/folding.rs:297-451, recursive/folding/src/lib.rs:23-90 */
    qm31 first[3][MAX_QUERIES];
    {
        uint32_t flat[3 * 2 * MAX_QUERIES * 4]; uint64_t nflat = 0, wi = 0;
        uint8_t has_data[32] = {0};
        for (uint32_t gi = 0; gi < n_sizes; gi++) {
            uint32_t l = sizes[gi];
            has_data[l] = 1;
            uint32_t ps[2 * MAX_QUERIES], np = 0;
            for (uint32_t j = 0; j < nq; j++) { uint32_t ql = qM[j] >> (M - l); ps[np++] = ql; ps[np++] = ql ^ 1; }
            np = sort_dedup(ps, np);
            qm31 vals[2 * MAX_QUERIES];
            for (uint32_t i = 0; i < np; i++) {
                int found = 0;
                for (uint32_t j = 0; j < nq && !found; j++)
                    if ((qM[j] >> (M - l)) == ps[i]) { vals[i] = answers[gi][j]; found = 1; }
                if (!found) {
                    if (wi >= v->first.n_witness) FAIL(RSV_R_FRI_FIRST);
                    const uint32_t* w = v->first.fri_witness + 4 * wi++;
                    vals[i] = q_mk(w[0], w[1], w[2], w[3]);
                }
                flat[nflat++] = vals[i].a.a; flat[nflat++] = vals[i].a.b;
                flat[nflat++] = vals[i].b.a; flat[nflat++] = vals[i].b.b;
            }
            /* fold circle -> line: recursive/folding/src/lib.rs:57-90 */
            for (uint32_t j = 0; j < nq; j++) {
                uint32_t ql = qM[j] >> (M - l);
                qm31 self = answers[gi][j], sib = Q_ZERO;
                for (uint32_t i = 0; i < np; i++) if (ps[i] == (ql ^ 1)) sib = vals[i];
                qm31 lv = (ql & 1) ? sib : self, rv = (ql & 1) ? self : sib;
                cpoint pt = domain_point(l, ql & ~1u);
                qm31 f = q_add(q_add(lv, rv), q_mul(q_mul_m(q_sub(lv, rv), m_inv(pt.y)), t->fri_alphas[M - l]));
                first[gi][j] = f;
            }
        }
        memcpy(probe->first_folded, first, sizeof first);
        if (wi != v->first.n_witness) FAIL(RSV_R_FRI_FIRST); /* folding.rs:367 */
        if (probe->pair_records) { g_pair_record = &probe->pair_records[0]; probe->n_inner = v->n_inner; probe->M = M; probe->nq = nq; memcpy(probe->qM, qM, 4 * nq); }
        int first_ok = verify_pair_tree(qM, nq, M, has_data, flat, nflat, &v->first.decommit, v->first.commitment);
        g_pair_record = NULL;
        if (!first_ok) FAIL(RSV_R_FRI_FIRST);
    }
    /* FRI inner layers: components/hints/src/folding.rs:460-566, recursive/folding/src/lib.rs:120-192 */
    qm31 folded[MAX_QUERIES];
    for (uint32_t j = 0; j < nq; j++) folded[j] = Q_ZERO;
    uint32_t l = M;
    for (uint32_t i = 0; i < v->n_inner; i++) {
        for (uint32_t gi = 0; gi < n_sizes; gi++)
            if (sizes[gi] == l) {
                qm31 a2 = q_mul(t->fri_alphas[i], t->fri_alphas[i]);
                for (uint32_t j = 0; j < nq; j++) folded[j] = q_add(q_mul(a2, folded[j]), first[gi][j]);
            }
        for (uint32_t j = 0; j < nq; j++) probe->inner_in[i][j] = folded[j];
        l -= 1;
        const fri_layer_view* L = &v->inner[i];
        uint32_t ps[2 * MAX_QUERIES], np = 0, pq[MAX_QUERIES];
        for (uint32_t j = 0; j < nq; j++) { pq[j] = qM[j] >> (M - l); ps[np++] = pq[j]; ps[np++] = pq[j] ^ 1; }
        np = sort_dedup(ps, np);
        qm31 vals[2 * MAX_QUERIES]; uint64_t wi = 0;
        uint32_t flat[2 * MAX_QUERIES * 4]; uint64_t nflat = 0;
        for (uint32_t k = 0; k < np; k++) {
            int found = 0;
            for (uint32_t j = 0; j < nq && !found; j++) if (pq[j] == ps[k]) { vals[k] = folded[j]; found = 1; }
            if (!found) {
                if (wi >= L->n_witness) FAIL(RSV_R_FRI_INNER);
                const uint32_t* w = L->fri_witness + 4 * wi++;
                vals[k] = q_mk(w[0], w[1], w[2], w[3]);
            }
            flat[nflat++] = vals[k].a.a; flat[nflat++] = vals[k].a.b;
            flat[nflat++] = vals[k].b.a; flat[nflat++] = vals[k].b.b;
        }
        if (wi != L->n_witness) FAIL(RSV_R_FRI_INNER); /* folding.rs:558 */
        uint8_t has_data[32] = {0};
        has_data[l] = 1;
        if (probe->pair_records) g_pair_record = &probe->pair_records[1 + i];
        int inner_ok = verify_pair_tree(pq, nq, l, has_data, flat, nflat, &L->decommit, L->commitment);
        g_pair_record = NULL;
        if (!inner_ok) FAIL(RSV_R_FRI_INNER);
        for (uint32_t j = 0; j < nq; j++) {
            qm31 self = folded[j], sib = Q_ZERO;
            for (uint32_t k = 0; k < np; k++) if (ps[k] == (pq[j] ^ 1)) sib = vals[k];
            qm31 lv = (pq[j] & 1) ? sib : self, rv = (pq[j] & 1) ? self : sib;
            m31 x = half_odds_at(l, bit_reverse(pq[j] & ~1u, l)).x;
            folded[j] = q_add(q_add(lv, rv), q_mul(q_mul_m(q_sub(lv, rv), m_inv(x)), t->fri_alphas[i + 1]));
        }
    }
    /* last layer: components/hints/src/folding.rs:569-595, recursive/folding/src/lib.rs:194-204 */
    {
        uint32_t ll = l - 1;
        uint32_t log_n = 0;
        while ((1ull << log_n) < v->n_last) log_n++;
        for (uint32_t j = 0; j < nq; j++) {
            probe->last_value[j] = folded[j];
            uint32_t idx = (qM[j] >> (M - l)) >> 1;
            m31 x = half_odds_at(ll, bit_reverse(idx, ll)).x, d[32];
            for (uint32_t k = 0; k < log_n; k++) { d[k] = x; x = m_sub(m_add(m_mul(x, x), m_mul(x, x)), 1); }
            probe->last_eval[j] = line_fold(v->last_coeffs, v->n_last, d);
            if (!q_eq(probe->last_eval[j], folded[j])) reason = RSV_R_FRI_LAST;
        }
    }
done:
    free(groups); free(cols); free(t); free(v);
    return reason;
#undef FAIL
}

/* cfg is REQUIRED, as in the reference (FiatShamirHints::new(&proof, config, ..), components/hints/src/fiat_shamir.rs:69-74):
 * proof i must carry cfgs[cfg_of ? cfg_of[i] : 0]; an index out of range rejects the proof (PARSE). */
int rsvo_verify_batch(const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg,
                      const rsv_public_input* pi, size_t n_pi, uint8_t* accept, uint8_t* reason) {
    if (n && (!blob || !offsets || !accept)) return RSV_E_NULL;
    if (n_pi && !pi) return RSV_E_NULL;
    if (!cfg || !cfg->cfgs) return RSV_E_NULL;
    if (cfg->n_cfgs < 1 || cfg->n_cfgs > RSV_MAX_CFGS) return RSV_E_SIZE;
    for (size_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return RSV_E_SIZE;
    for (size_t i = 0; i < n_pi; i++) if (!canonical(pi[i].value, 4)) return RSV_E_RANGE;
    query_probe* scratch = calloc(1, sizeof *scratch);
    for (size_t i = 0; i < n; i++) {
        const uint32_t ci = cfg->cfg_of ? cfg->cfg_of[i] : 0u;
        uint8_t r = ci < cfg->n_cfgs
                        ? verify_one(blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &cfg->cfgs[ci], pi, n_pi, scratch)
                        : (uint8_t)RSV_R_PARSE;
        accept[i] = r == RSV_R_OK;
        if (reason) reason[i] = r;
    }
    free(scratch);
    return RSV_OK;
}

int rsvo_query_values(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi,
                      uint32_t* out, size_t cap) {
    if (!proof || !out) return RSV_E_NULL;
    query_probe* pr = calloc(1, sizeof *pr);
    proof_view* v = malloc(sizeof *v);
    int rc;
    if (!parse_proof(proof, len, NULL, v)) { rc = RSV_E_SIZE; goto done; }
    uint8_t r = verify_one(proof, len, NULL, pi, n_pi, pr);
    if (r != RSV_R_OK && r != RSV_R_FRI_LAST) { rc = RSV_E_SIZE; goto done; }
    uint32_t nq = v->cfg.n_queries;
    size_t need = (size_t)nq * (pr->n_sizes + 1) * 4;
    if (cap < need) { rc = RSV_E_CAP; goto done; }
    size_t o = 0;
    for (uint32_t j = 0; j < nq; j++) {
        for (uint32_t g = 0; g <= pr->n_sizes; g++) {
            qm31 q = g < pr->n_sizes ? pr->answers[g][j] : pr->last_value[j];
            out[o++] = q.a.a; out[o++] = q.a.b; out[o++] = q.b.a; out[o++] = q.b.b;
        }
    }
    rc = (int)o;
done:
    free(pr); free(v);
    return rc;
}

/* The layout of rsv_hints_out::d_query_values for one proof: out [nq][4 * (8 + n_inner)], per query (transcript
 * order) answers[3] | first-layer folds[3] | value entering inner layer i | value entering the last-layer check |
 * last-layer polynomial at the query's point. */
int rsvo_query_dump(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out, size_t cap,
                    uint32_t* n_inner_out, uint32_t* nq_out) {
    if (!proof || !out) return RSV_E_NULL;
    query_probe* pr = calloc(1, sizeof *pr);
    proof_view* v = malloc(sizeof *v);
    int rc = RSV_OK;
    if (!parse_proof(proof, len, NULL, v)) { rc = RSV_E_SIZE; goto done; }
    uint8_t r = verify_one(proof, len, NULL, pi, n_pi, pr);
    if (r != RSV_R_OK && r != RSV_R_FRI_LAST) { rc = RSV_E_SIZE; goto done; }
    const uint32_t nq = v->cfg.n_queries, ni = v->n_inner;
    const size_t stride = 4 * (8 + (size_t)ni);
    if (cap < nq * stride) { rc = RSV_E_CAP; goto done; }
    memset(out, 0, 4 * nq * stride);
    for (uint32_t j = 0; j < nq; j++) {
        uint32_t* o = out + j * stride;
#define PUT(off, q) do { o[off] = (q).a.a; o[(off) + 1] = (q).a.b; o[(off) + 2] = (q).b.a; o[(off) + 3] = (q).b.b; } while (0)
        for (uint32_t g = 0; g < pr->n_sizes; g++) { PUT(4 * g, pr->answers[g][j]); PUT(12 + 4 * g, pr->first_folded[g][j]); }
        for (uint32_t i = 0; i < ni; i++) PUT(24 + 4 * i, pr->inner_in[i][j]);
        PUT(24 + 4 * ni, pr->last_value[j]);
        PUT(28 + 4 * ni, pr->last_eval[j]);
#undef PUT
    }
    if (n_inner_out) *n_inner_out = ni;
    if (nq_out) *nq_out = nq;
done:
    free(pr); free(v);
    return rc;
}

/* FirstLayerHints::folded_evals_by_column (components/hints/src/folding.rs:343-360): out [3][nq][4], the circle->line
 * fold of every query's first-layer pair at each column log size (descending), transcript query order. */
int rsvo_fri_folded(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out, size_t cap,
                    uint32_t* n_sizes, uint32_t* n_queries) {
    if (!proof || !out || !n_sizes || !n_queries) return RSV_E_NULL;
    query_probe* pr = calloc(1, sizeof *pr);
    proof_view* v = malloc(sizeof *v);
    int rc = RSV_OK;
    if (!parse_proof(proof, len, NULL, v)) { rc = RSV_E_SIZE; goto done; }
    if (verify_one(proof, len, NULL, pi, n_pi, pr) != RSV_R_OK) { rc = RSV_E_SIZE; goto done; }
    const uint32_t nq = v->cfg.n_queries;
    if (cap < (size_t)3 * nq * 4) { rc = RSV_E_CAP; goto done; }
    *n_sizes = pr->n_sizes; *n_queries = nq;
    for (uint32_t g = 0; g < 3; g++)
        for (uint32_t j = 0; j < nq; j++) {
            qm31 q = g < pr->n_sizes ? pr->first_folded[g][j] : Q_ZERO;
            uint32_t* o = out + ((size_t)g * nq + j) * 4;
            o[0] = q.a.a; o[1] = q.a.b; o[2] = q.b.a; o[3] = q.b.b;
        }
done:
    free(pr); free(v);
    return rc;
}

/* the per-query authentication paths SinglePathMerkleProof::from_stwo_proof
 * (components/hints/src/decommit.rs:144-183) cherry-picks from the batched walk, transcript query order.
 * sib layout [4][nq][M][8] (level k above the leaf at index k), pos [4][nq], depth4 [4]. */
int rsvo_trace_paths(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* sib,
                     size_t cap, uint32_t* pos, uint32_t* depth4, uint32_t* n_queries) {
    if (!proof || !sib || !pos || !depth4 || !n_queries) return RSV_E_NULL;
    query_probe* pr = calloc(1, sizeof *pr);
    pr->records = calloc(4, sizeof(tree_record));
    int rc = RSV_OK;
    uint8_t r = verify_one(proof, len, NULL, pi, n_pi, pr);
    if (r != RSV_R_OK) { rc = RSV_E_SIZE; goto done; }
    const uint32_t nq = pr->nq, M = pr->M;
    if (cap < (size_t)4 * nq * M * 8) { rc = RSV_E_CAP; goto done; }
    *n_queries = nq;
    for (int t = 0; t < 4; t++) {
        const tree_record* rec = &pr->records[t];
        const uint32_t mx = pr->maxlog[t];
        depth4[t] = mx;
        for (uint32_t i = 0; i < nq; i++) {
            uint32_t cur = pr->qM[i] >> (M - mx);
            pos[t * nq + i] = cur;
            for (uint32_t k = 0; k < mx; k++) {
                const node* sn = find_node(rec->nodes[mx - k], rec->n[mx - k], cur ^ 1);
                if (!sn) { rc = RSV_E_SIZE; goto done; }
                memcpy(sib + (((size_t)t * nq + i) * M + k) * 8, sn->h, 32);
                cur >>= 1;
            }
        }
    }
done:
    free(pr->records); free(pr);
    return rc;
}

/* SinglePathMerkleProof::columns for the same paths: cols [4][nq][64], per query the columns at the leaf level
 * followed by the columns at the lower column log size (trees 0..2 when lp + 1 != lq + 2... i.e. A != B). */
int rsvo_trace_cols(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* cols, size_t cap,
                    uint32_t* n_queries) {
    if (!proof || !cols || !n_queries) return RSV_E_NULL;
    if (cap < (size_t)4 * MAX_QUERIES * 64) return RSV_E_CAP;
    query_probe* pr = calloc(1, sizeof *pr);
    pr->trace_cols = cols;
    uint8_t r = verify_one(proof, len, NULL, pi, n_pi, pr);
    *n_queries = pr->nq;
    free(pr);
    return r == RSV_R_OK ? RSV_OK : RSV_E_SIZE;
}

/* SURVEY 8f.1, pair trees: what SinglePairMerkleProof::from_stwo_proof (components/hints/src/folding.rs:214-287)
 * cherry-picks, transcript query order.  Tree s = 0 is the FRI first layer (leaf level M), s = 1 + i inner layer i
 * (leaf level M - 1 - i).  For the tree of depth d:
 *   sib  [(s*nq + q)*M + k]      k = 0..d-2: sibling_hashes[k], i.e. at level d-1-k the sibling's hash, or — where
 *                                that level carries a column — the hash of the sibling's children
 *   cols [((s*nq + q)*3 + c)*8]  c-th data level from the top (first layer: up to 3, inner: 1): self value | sibling value */
int rsvo_fri_paths(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* sib,
                   size_t cap, uint32_t* cols, uint32_t* n_trees, uint32_t* n_queries) {
    if (!proof || !sib || !cols || !n_trees || !n_queries) return RSV_E_NULL;
    query_probe* pr = calloc(1, sizeof *pr);
    pr->pair_records = calloc(1 + MAX_LAYERS, sizeof(pair_record));
    int rc = RSV_OK;
    uint8_t r = verify_one(proof, len, NULL, pi, n_pi, pr);
    if (r != RSV_R_OK) { rc = RSV_E_SIZE; goto done; }
    const uint32_t nq = pr->nq, M = pr->M, nt = 1 + pr->n_inner;
    if (cap < (size_t)nt * nq * M * 8) { rc = RSV_E_CAP; goto done; }
    *n_trees = nt; *n_queries = nq;
    for (uint32_t s2 = 0; s2 < nt; s2++) {
        const pair_record* rec = &pr->pair_records[s2];
        const uint32_t d = s2 == 0 ? M : M - s2;
        for (uint32_t i = 0; i < nq; i++) {
            uint32_t cur = pr->qM[i] >> (M - d), c = 0, k = 0;
            for (uint32_t l = d; l >= 1; l--) {
                if (rec->nc[l]) {
                    const m31 *sv = NULL, *bv = NULL, *kids = NULL;
                    for (uint32_t t = 0; t < rec->nc[l]; t++) {
                        if (rec->cand[l][t].pos == cur) sv = rec->cand[l][t].val;
                        if (rec->cand[l][t].pos == (cur ^ 1)) { bv = rec->cand[l][t].val; kids = rec->cand[l][t].kids; }
                    }
                    if (!sv || !bv) { rc = RSV_E_SIZE; goto done; }
                    uint32_t* o = cols + (((size_t)s2 * nq + i) * 3 + c) * 8;
                    memcpy(o, sv, 16); memcpy(o + 4, bv, 16);
                    c++;
                    if (l != d) { memcpy(sib + (((size_t)s2 * nq + i) * M + k) * 8, kids, 32); k++; }
                } else {
                    const node* sn = find_node(rec->nodes[l], rec->n[l], cur ^ 1);
                    if (!sn) { rc = RSV_E_SIZE; goto done; }
                    memcpy(sib + (((size_t)s2 * nq + i) * M + k) * 8, sn->h, 32); k++;
                }
                cur >>= 1;
            }
        }
    }
done:
    free(pr->pair_records); free(pr);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY 8f.1, second half: the VALUE side of the recursion circuit's Poseidon accelerator — PoseidonFlow
 * (constraint_system/src/plonk_with_poseidon.rs:36,117-128): one record per Poseidon2HalfVar::permute invocation
 * (primitives/poseidon31/src/lib.rs:282-423), in the order the circuit that verifies this proof makes them
 * (examples/multi-proofs/src/main.rs:69-139):
 *   FiatShamirResults::compute        every channel operation                  fiat_shamir/src/lib.rs:44-130
 *   AnswerResults::compute            tree 0 for every query in TRANSCRIPT order, then trees 1, 2, 3;
 *                                     one SinglePathMerkleProofVar::verify each   answer/src/lib.rs:214-258,
 *                                                                                 data_structures/src/lib.rs:315-354
 *   FoldingResults::compute           the first-layer SinglePairMerkleProofVar::verify of every query, then per inner
 *                                     layer one per query                      folding/src/lib.rs:23-33,186-189,
 *                                                                              data_structures/src/lib.rs:400-464
 * (CompositionCheck and the quotient / fold arithmetic invoke no permutation.)  The restatement below runs exactly
 * those per-path verifications — on the per-query paths rsvo_trace_paths / rsvo_trace_cols / rsvo_fri_paths extract,
 * i.e. what SinglePathMerkleProof / SinglePairMerkleProof::from_stwo_proof hand the circuit — with the recorder of
 * half_permute switched on, so the records come out in invocation order by construction.
 * out: [count][33] = left8 | right8 | out_rate8 | out_cap8 | swap.  Returns RSV_OK and *count (also when cap is too
 * small: then only the first cap records are written and the return value is RSV_E_CAP).
 * Pinned by the reference only through the COUNT (tests/test_oracle.py: the padded flow of level K's verification is
 * the Poseidon trace of level K+1, whose log size is written in that fixture's header); beyond that: parity unpinned. */
static void flow_sponge_rate(const m31* cols, size_t n, m31* out) { /* hash_m31_columns_get_rate, merkle/src/lib.rs:50-91 */
    static const m31 zero8[8] = {0};
    m31 d[8];
    sponge_capacity(cols, n, d);
    half_permute(zero8, d, 0, out, NULL);
}
static void flow_qm31_capacity(const m31* v4, m31* d) { /* hash_qm31_columns_get_capacity(&[v, 0]), merkle/src/lib.rs:99-139 */
    static const m31 zero8[8] = {0};
    m31 chunk[8] = {v4[0], v4[1], v4[2], v4[3], 0, 0, 0, 0};
    half_permute(chunk, zero8, 0, NULL, d);
}
static void flow_qm31_rate(const m31* v4, m31* out) { /* hash_qm31_columns_get_rate(&[v, 0]), merkle/src/lib.rs:93-97 */
    static const m31 zero8[8] = {0};
    m31 d[8];
    flow_qm31_capacity(v4, d);
    half_permute(zero8, d, 0, out, NULL);
}

int rsvo_poseidon_flow(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out,
                       size_t cap, size_t* count) {
    if (!proof || !count || (cap && !out)) return RSV_E_NULL;
    int rc = RSV_OK;
    proof_view* v = malloc(sizeof *v);
    transcript* t = malloc(sizeof *t);
    uint32_t *tsib = NULL, *tcols = NULL, *fsib = NULL, *fcols = NULL;
    flow_recorder fr = {out, cap, 0};
    if (!parse_proof(proof, len, NULL, v)) { rc = RSV_E_SIZE; goto done; }
    const uint32_t nq = v->cfg.n_queries, M = v->M, A = v->A, B = v->B, nt = 1 + v->n_inner;
    uint32_t tpos[4 * MAX_QUERIES], depth4[4], nq2 = 0, nt2 = 0;
    tsib = malloc((size_t)4 * nq * M * 8 * 4); tcols = calloc((size_t)4 * MAX_QUERIES * 64, 4);
    fsib = calloc((size_t)nt * nq * M * 8, 4); fcols = calloc((size_t)nt * nq * 3 * 8, 4);
    /* the per-query paths (these calls verify the proof; a proof that does not verify has no flow) */
    if (rsvo_trace_paths(proof, len, pi, n_pi, tsib, (size_t)4 * nq * M * 8, tpos, depth4, &nq2) != RSV_OK || nq2 != nq ||
        rsvo_trace_cols(proof, len, pi, n_pi, tcols, (size_t)4 * MAX_QUERIES * 64, &nq2) != RSV_OK ||
        rsvo_fri_paths(proof, len, pi, n_pi, fsib, (size_t)nt * nq * M * 8, fcols, &nt2, &nq2) != RSV_OK || nt2 != nt) {
        rc = RSV_E_SIZE; goto done;
    }
    g_flow = &fr;
    run_transcript(v, t);
    /* AnswerResults::compute: four trees, every query in transcript order (answer/src/lib.rs:214-258) */
    for (int tr = 0; tr < 4; tr++) {
        static const uint32_t plonk_cols[3] = {10, 12, 8}, poseidon_cols[3] = {40, 48, 8};
        uint32_t ncols_at[32] = {0};
        if (tr < 3) { ncols_at[A] += plonk_cols[tr]; ncols_at[B] += poseidon_cols[tr]; } else ncols_at[M] = 8;
        const uint32_t depth = depth4[tr];
        for (uint32_t i = 0; i < nq; i++) {
            /* SinglePathMerkleProofVar::verify (data_structures/src/lib.rs:315-354) */
            const m31* cols = tcols + ((size_t)tr * nq + i) * 64;
            const uint32_t query = tpos[tr * nq + i];
            m31 cur[8];
            flow_sponge_rate(cols, ncols_at[depth], cur);
            cols += ncols_at[depth];
            for (uint32_t k = 0; k < depth; k++) {
                const uint32_t h = depth - k - 1;
                const m31* sib = tsib + (((size_t)tr * nq + i) * M + k) * 8;
                const int bit = (query >> k) & 1;
                if (ncols_at[h]) { /* hash_tree_with_column_hash_with_swap (merkle/src/lib.rs:32-41): column hash first */
                    m31 ch[8], tmp[8];
                    sponge_capacity(cols, ncols_at[h], ch);
                    cols += ncols_at[h];
                    half_permute(cur, sib, bit, tmp, NULL);
                    half_permute(tmp, ch, 0, cur, NULL);
                } else half_permute(cur, sib, bit, cur, NULL); /* hash_tree_with_swap (merkle/src/lib.rs:22-30) */
            }
            if (memcmp(cur, v->commitments[tr], 32) != 0) rc = RSV_E_SIZE;
        }
    }
    /* FoldingResults::compute: first layer for every query, then every inner layer for every query */
    for (uint32_t s2 = 0; s2 < nt; s2++) {
        const uint32_t depth = s2 == 0 ? M : M - s2;
        uint8_t data[32] = {0};
        if (s2 == 0) { data[M] = 1; data[A] = 1; data[B] = 1; } else data[depth] = 1;
        const uint32_t* root = s2 == 0 ? v->first.commitment : v->inner[s2 - 1].commitment;
        for (uint32_t i = 0; i < nq; i++) {
            /* SinglePairMerkleProofVar::verify (data_structures/src/lib.rs:400-464) */
            const uint32_t query = (t->raw_queries[i] & ((1u << M) - 1)) >> (M - depth);
            const m31* cl = fcols + ((size_t)s2 * nq + i) * 3 * 8; /* c-th data level from the top: self | sibling */
            const m31* sh = fsib + ((size_t)s2 * nq + i) * M * 8;
            m31 self[8], sibling[8];
            uint32_t c = 0;
            flow_qm31_rate(cl, self);
            flow_qm31_rate(cl + 4, sibling);
            c++;
            for (uint32_t k = 0; k < depth; k++) {
                const uint32_t h = depth - k - 1;
                const int bit = (query >> k) & 1;
                if (!data[h] || h == 0) {
                    half_permute(self, sibling, bit, self, NULL);
                    if (k != depth - 1) memcpy(sibling, sh + 8 * k, 32);
                } else {
                    m31 sc[8], bc[8], tmp[8];
                    flow_qm31_capacity(cl + 8 * c, sc);
                    flow_qm31_capacity(cl + 8 * c + 4, bc);
                    c++;
                    half_permute(self, sibling, bit, tmp, NULL);
                    half_permute(tmp, sc, 0, self, NULL);
                    half_permute(sh + 8 * k, bc, 0, sibling, NULL);
                }
            }
            if (memcmp(self, root, 32) != 0) rc = RSV_E_SIZE;
        }
    }
    g_flow = NULL;
    *count = fr.n;
    if (rc == RSV_OK && fr.n > cap) rc = RSV_E_CAP;
done:
    g_flow = NULL;
    free(tsib); free(tcols); free(fsib); free(fcols); free(t); free(v);
    return rc;
}
