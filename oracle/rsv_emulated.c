/*
 * rsv_emulated.c — TEST INFRASTRUCTURE ONLY (part of librsv_oracle.so, see rsv_oracle.h).
 *
 * CPU restatement of the reference's "emulated" Poseidon2: the permutation written as gates of the
 * Plonk-without-Poseidon constraint system, together with just enough of that constraint system to replay the
 * reference's own test of it.
 *
 *   constraint system   constraint_system/src/plonk_without_poseidon.rs:32-91   four fixed variables / rows
 *                       :93-260  insert_gate, do_m4_gate, do_pow5m4_gate, do_pow5_gate, do_hadamard, do_grandsum_gate
 *                       :274-305 add, mul, mul_constant          :307-407 new_m31, new_qm31 (three allocation modes)
 *                       :410-598 check_arithmetics (the gate equations, restated in ecs_check)
 *   variables           primitives/fields/src/m31.rs:35-61, qm31.rs:39-73 (constants are cached by value),
 *                       qm31.rs:129-164, 182-193, 237-249 (+, -, * allocate), :255-265 (from_m31)
 *   the gadget          primitives/poseidon31/src/emulated.rs:12-76 (m4, 16x16 MDS, pow5m4, pow5), :80-221
 *
 * Parity status: PINNED by the reference's own test (emulated.rs:236-275): 16 witness words 0..15, three permutes
 * (no swap, Some((false, 0)), Some((true, 1)) with the halves exchanged) must all give the known-answer state, and
 * every row must satisfy check_arithmetics.  tests/test_oracle.py replays exactly that.
 */
#include <stdlib.h>
#include <string.h>
#include "rsv_oracle.h"

#define P 0x7fffffffu
typedef uint32_t m31;
typedef struct { m31 v[4]; } q31; /* (v0 + v1 i) + (v2 + v3 i) u */

static m31 f_add(m31 a, m31 b) { uint32_t s = a + b; return s >= P ? s - P : s; }
static m31 f_sub(m31 a, m31 b) { return a >= b ? a - b : a + P - b; }
static m31 f_mul(m31 a, m31 b) { return (m31)(((uint64_t)a * b) % P); }
static m31 f_pow4(m31 a) { m31 s = f_mul(a, a); return f_mul(s, s); }

static q31 q_of(m31 a, m31 b, m31 c, m31 d) { q31 r = {{a, b, c, d}}; return r; }
static q31 q_add(q31 a, q31 b) { q31 r; for (int i = 0; i < 4; i++) r.v[i] = f_add(a.v[i], b.v[i]); return r; }
static q31 q_scale(q31 a, m31 k) { q31 r; for (int i = 0; i < 4; i++) r.v[i] = f_mul(a.v[i], k); return r; }
/* (a + b u)(c + d u) = ac + bd (2 + i) + (ad + bc) u over CM31 */
static void c_mul(const m31* a, const m31* b, m31* o) {
    m31 re = f_sub(f_mul(a[0], b[0]), f_mul(a[1], b[1])), im = f_add(f_mul(a[0], b[1]), f_mul(a[1], b[0]));
    o[0] = re; o[1] = im;
}
static q31 q_mul(q31 x, q31 y) {
    m31 ac[2], bd[2], ad[2], bc[2];
    c_mul(&x.v[0], &y.v[0], ac); c_mul(&x.v[2], &y.v[2], bd); c_mul(&x.v[0], &y.v[2], ad); c_mul(&x.v[2], &y.v[0], bc);
    /* bd * (2 + i) = (2 bd0 - bd1) + (2 bd1 + bd0) i */
    m31 r0 = f_sub(f_add(bd[0], bd[0]), bd[1]), r1 = f_add(f_add(bd[1], bd[1]), bd[0]);
    return q_of(f_add(ac[0], r0), f_add(ac[1], r1), f_add(ad[0], bc[0]), f_add(ad[1], bc[1]));
}
static int q_eq(q31 a, q31 b) { return memcmp(a.v, b.v, sizeof a.v) == 0; }

/* M4 of the Poseidon2 paper as both the gates and check_arithmetics spell it (plonk_without_poseidon.rs:122-129, :436-452) */
static q31 q_m4(q31 x) {
    m31 t0 = f_add(x.v[0], x.v[1]), t1 = f_add(x.v[2], x.v[3]);
    m31 t2 = f_add(f_add(x.v[1], x.v[1]), t1), t3 = f_add(f_add(x.v[3], x.v[3]), t0);
    m31 t12 = f_add(t1, t1), t02 = f_add(t0, t0);
    m31 t4 = f_add(f_add(t12, t12), t3), t5 = f_add(f_add(t02, t02), t2);
    return q_of(f_add(t3, t5), t5, f_add(t2, t4), t4);
}
static q31 q_hadamard(q31 a, q31 b) { q31 r; for (int i = 0; i < 4; i++) r.v[i] = f_mul(a.v[i], b.v[i]); return r; }
static q31 q_pow4(q31 a) { q31 r; for (int i = 0; i < 4; i++) r.v[i] = f_pow4(a.v[i]); return r; }
static q31 q_grand(q31 a, q31 b) {
    m31 s = 0;
    for (int i = 0; i < 4; i++) s = f_add(s, a.v[i]);
    for (int i = 0; i < 4; i++) s = f_add(s, b.v[i]);
    return q_of(s, s, s, s);
}

/* ------------------------------------------------------------------------------------------ constraint system */
typedef struct { uint32_t a, b, c; m31 op1, op2, op3, op4; } row;
typedef struct { int is_q; q31 value; uint32_t var; } cached;
struct rsvo_ecs {
    q31* vars; uint8_t* kind; size_t n_vars, cap_vars;
    row* rows; size_t n_rows, cap_rows;
    cached* cache; size_t n_cache, cap_cache;
    int in_constant; /* > 0 while a constant is being allocated: its helper variables are tagged RSVO_VAR_CONSTANT */
};

static uint32_t push_var(rsvo_ecs* cs, q31 v, uint8_t kind) {
    if (cs->n_vars == cs->cap_vars) {
        cs->cap_vars = cs->cap_vars ? 2 * cs->cap_vars : 1024;
        cs->vars = realloc(cs->vars, cs->cap_vars * sizeof(q31));
        cs->kind = realloc(cs->kind, cs->cap_vars);
    }
    cs->vars[cs->n_vars] = v;
    cs->kind[cs->n_vars] = cs->in_constant ? RSVO_VAR_CONSTANT : kind;
    return (uint32_t)cs->n_vars++;
}
static void push_row(rsvo_ecs* cs, uint32_t a, uint32_t b, uint32_t c, m31 op1, m31 op2, m31 op3, m31 op4) {
    if (cs->n_rows == cs->cap_rows) {
        cs->cap_rows = cs->cap_rows ? 2 * cs->cap_rows : 1024;
        cs->rows = realloc(cs->rows, cs->cap_rows * sizeof(row));
    }
    row r = {a, b, c, op1, op2, op3, op4};
    cs->rows[cs->n_rows++] = r;
}

/* :32-91 */
rsvo_ecs* rsvo_ecs_new(void) {
    rsvo_ecs* cs = calloc(1, sizeof *cs);
    push_var(cs, q_of(0, 0, 0, 0), RSVO_VAR_FIXED);
    push_var(cs, q_of(1, 0, 0, 0), RSVO_VAR_FIXED);
    push_var(cs, q_of(0, 1, 0, 0), RSVO_VAR_FIXED);
    push_var(cs, q_of(0, 0, 1, 0), RSVO_VAR_FIXED);
    push_row(cs, 0, 0, 0, 1, 0, 0, 0);
    push_row(cs, 1, 0, 1, 1, 0, 0, 0);
    push_row(cs, 2, 0, 2, 1, 0, 0, 0);
    push_row(cs, 3, 0, 3, 1, 0, 0, 0);
    return cs;
}
void rsvo_ecs_free(rsvo_ecs* cs) {
    if (!cs) return;
    free(cs->vars); free(cs->kind); free(cs->rows); free(cs->cache); free(cs);
}
size_t rsvo_ecs_n_vars(const rsvo_ecs* cs) { return cs->n_vars; }
size_t rsvo_ecs_n_rows(const rsvo_ecs* cs) { return cs->n_rows; }
void rsvo_ecs_export(const rsvo_ecs* cs, uint32_t* vars4, uint8_t* kind, uint32_t* rows7) {
    if (vars4) memcpy(vars4, cs->vars, cs->n_vars * sizeof(q31));
    if (kind) memcpy(kind, cs->kind, cs->n_vars);
    if (rows7) memcpy(rows7, cs->rows, cs->n_rows * sizeof(row));
}
/* overwrite variable values (tests substitute the GPU's rows and re-run the gate equations) */
int rsvo_ecs_set_vars(rsvo_ecs* cs, size_t first, const uint32_t* vars4, size_t n) {
    if (first + n > cs->n_vars) return RSV_E_RANGE;
    memcpy(cs->vars + first, vars4, n * sizeof(q31));
    return RSV_OK;
}

/* :274-305 */
static uint32_t cs_add(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_add(cs->vars[a], cs->vars[b]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 0, 0, 0);
    return c;
}
static uint32_t cs_mul(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_mul(cs->vars[a], cs->vars[b]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 0, 0, 0, 0);
    return c;
}
static uint32_t cs_mul_constant(rsvo_ecs* cs, uint32_t a, m31 k) {
    uint32_t c = push_var(cs, q_scale(cs->vars[a], k), RSVO_VAR_GATE);
    push_row(cs, a, 0, c, k, 0, 0, 0);
    return c;
}
/* :113-260 */
static uint32_t cs_m4(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_m4(cs->vars[a]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 0, 1, 0);
    return c;
}
static uint32_t cs_pow5m4(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_m4(q_hadamard(cs->vars[a], cs->vars[b])), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 1, 1, 0);
    return c;
}
static uint32_t cs_pow5(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_hadamard(cs->vars[a], cs->vars[b]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 1, 0, 1);
    return c;
}
static uint32_t cs_hadamard(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_hadamard(cs->vars[a], cs->vars[b]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 0, 0, 1);
    return c;
}
static uint32_t cs_grandsum(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t c = push_var(cs, q_grand(cs->vars[a], cs->vars[b]), RSVO_VAR_GATE);
    push_row(cs, a, b, c, 1, 0, 1, 1);
    return c;
}
/* :307-345: mode 0 = witness, 1 = constant (public inputs are not needed here) */
static uint32_t cs_new_m31(rsvo_ecs* cs, m31 v, int constant) {
    uint32_t c = push_var(cs, q_of(v, 0, 0, 0), RSVO_VAR_WITNESS);
    if (!constant) push_row(cs, c, 1, c, 1, 0, 0, 1);
    else push_row(cs, 1, 0, c, v, 0, 0, 0);
    return c;
}
/* :347-407 */
static uint32_t cs_new_qm31(rsvo_ecs* cs, q31 v, int constant) {
    uint32_t c = push_var(cs, v, RSVO_VAR_WITNESS);
    if (!constant) { push_row(cs, c, 0, c, 1, 0, 0, 0); return c; }
    uint32_t re0 = cs_new_m31(cs, v.v[0], 1), im0 = cs_new_m31(cs, v.v[1], 1);
    uint32_t re1 = cs_new_m31(cs, v.v[2], 1), im1 = cs_new_m31(cs, v.v[3], 1);
    uint32_t t = cs_mul(cs, im0, 2);
    uint32_t a = cs_add(cs, re0, t);
    t = cs_mul(cs, im1, 2);
    t = cs_add(cs, re1, t);
    uint32_t b = cs_mul(cs, t, 3);
    push_row(cs, a, b, c, 1, 0, 0, 0);
    return c;
}
uint32_t rsvo_ecs_new_witness_m31(rsvo_ecs* cs, uint32_t v) { return cs_new_m31(cs, v, 0); }
uint32_t rsvo_ecs_new_witness_qm31(rsvo_ecs* cs, const uint32_t* v4) { return cs_new_qm31(cs, q_of(v4[0], v4[1], v4[2], v4[3]), 0); }

/* cached constants: m31.rs:35-61, qm31.rs:39-73 (keys "m31 v" / "qm31 a,b,c,d" are distinct name spaces) */
static uint32_t cache_get(rsvo_ecs* cs, int is_q, q31 v) {
    for (size_t i = 0; i < cs->n_cache; i++)
        if (cs->cache[i].is_q == is_q && q_eq(cs->cache[i].value, v)) return cs->cache[i].var;
    return UINT32_MAX;
}
static void cache_put(rsvo_ecs* cs, int is_q, q31 v, uint32_t var) {
    if (cs->n_cache == cs->cap_cache) {
        cs->cap_cache = cs->cap_cache ? 2 * cs->cap_cache : 256;
        cs->cache = realloc(cs->cache, cs->cap_cache * sizeof(cached));
    }
    cached e = {is_q, v, var};
    cs->cache[cs->n_cache++] = e;
}
static uint32_t const_m31(rsvo_ecs* cs, m31 v) {
    if (v == 0) return 0;
    if (v == 1) return 1;
    q31 key = q_of(v, 0, 0, 0);
    uint32_t var = cache_get(cs, 0, key);
    if (var != UINT32_MAX) return var;
    cs->in_constant++;
    var = cs_new_m31(cs, v, 1);
    cs->in_constant--;
    cache_put(cs, 0, key, var);
    return var;
}
static uint32_t const_qm31(rsvo_ecs* cs, q31 v) {
    if (q_eq(v, q_of(0, 0, 0, 0))) return 0;
    if (q_eq(v, q_of(1, 0, 0, 0))) return 1;
    if (q_eq(v, q_of(0, 1, 0, 0))) return 2;
    if (q_eq(v, q_of(0, 0, 1, 0))) return 3;
    uint32_t var = cache_get(cs, 1, v);
    if (var != UINT32_MAX) return var;
    cs->in_constant++;
    var = cs_new_qm31(cs, v, 1);
    cs->in_constant--;
    cache_put(cs, 1, v, var);
    return var;
}

/* qm31.rs:255-265; Rust evaluates the nested calls left to right, innermost first */
uint32_t rsvo_ecs_qm31_from_m31(rsvo_ecs* cs, const uint32_t* m4) {
    uint32_t l = cs_add(cs, m4[0], cs_mul(cs, m4[1], 2));
    uint32_t r = cs_mul(cs, cs_add(cs, m4[2], cs_mul(cs, m4[3], 2)), 3);
    return cs_add(cs, l, r);
}

/* ------------------------------------------------------------------------------------------ the gadget */
/* qm31.rs:158-164, 237-249: a - b = a + (b * (-1)) */
static uint32_t v_sub(rsvo_ecs* cs, uint32_t a, uint32_t b) {
    uint32_t nb = cs_mul_constant(cs, b, P - 1);
    return cs_add(cs, a, nb);
}
/* emulated.rs:12-22 */
static uint32_t g_m4(rsvo_ecs* cs, uint32_t x) {
    uint32_t k = const_qm31(cs, q_of(1, 1, 1, 1));
    return cs_m4(cs, x, k);
}
/* emulated.rs:24-35 */
static void g_mds16(rsvo_ecs* cs, uint32_t* st) {
    uint32_t p[4];
    for (int i = 0; i < 4; i++) p[i] = g_m4(cs, st[i]);
    uint32_t t = cs_add(cs, p[0], p[1]);
    t = cs_add(cs, t, p[2]);
    t = cs_add(cs, t, p[3]);
    for (int i = 0; i < 4; i++) st[i] = cs_add(cs, p[i], t);
}
/* emulated.rs:37-61: the fourth powers are a fresh witness, the gate checks them and returns M4(x^5) */
static uint32_t g_pow5m4(rsvo_ecs* cs, uint32_t x) {
    uint32_t b = cs_new_qm31(cs, q_pow4(cs->vars[x]), 0);
    return cs_pow5m4(cs, x, b);
}
/* emulated.rs:63-78 */
static uint32_t g_pow5(rsvo_ecs* cs, uint32_t x) {
    uint32_t b = cs_new_qm31(cs, q_pow4(cs->vars[x]), 0);
    return cs_pow5(cs, x, b);
}
static void g_full_round(rsvo_ecs* cs, uint32_t* st, const m31* rc16) {
    for (int i = 0; i < 4; i++) st[i] = cs_add(cs, st[i], const_qm31(cs, q_of(rc16[4 * i], rc16[4 * i + 1], rc16[4 * i + 2], rc16[4 * i + 3])));
    for (int i = 0; i < 4; i++) st[i] = g_pow5m4(cs, st[i]);
    uint32_t t = cs_add(cs, st[0], st[1]);
    t = cs_add(cs, t, st[2]);
    t = cs_add(cs, t, st[3]);
    for (int i = 0; i < 4; i++) st[i] = cs_add(cs, st[i], t);
}

/* emulated.rs:80-221.  left / right: two QM31 variables each; swap_mode 0 = None, 1 = Some((bit value of
 * variable bit_var, bit_var)).  Returns the four output variables (left half first). */
int rsvo_ecs_permute_emulated(rsvo_ecs* cs, const uint32_t* left2, const uint32_t* right2, int swap_mode,
                              uint32_t bit_var, uint32_t* out4) {
    const m31* rc_first = rsvo_round_constants(0);
    const m31* rc_partial = rsvo_round_constants(1);
    const m31* rc_last = rsvo_round_constants(2);
    for (int i = 0; i < 2; i++)
        if (left2[i] >= cs->n_vars || right2[i] >= cs->n_vars) return RSV_E_RANGE;
    uint32_t st[4];
    if (swap_mode) {
        if (bit_var >= cs->n_vars) return RSV_E_RANGE;
        uint32_t rml[2], rmlb[2];
        for (int i = 0; i < 2; i++) rml[i] = v_sub(cs, right2[i], left2[i]);
        for (int i = 0; i < 2; i++) rmlb[i] = cs_mul(cs, rml[i], bit_var);
        for (int i = 0; i < 2; i++) st[i] = cs_add(cs, rmlb[i], left2[i]);
        for (int i = 0; i < 2; i++) st[2 + i] = v_sub(cs, right2[i], rmlb[i]);
    } else {
        st[0] = left2[0]; st[1] = left2[1]; st[2] = right2[0]; st[3] = right2[1];
    }
    g_mds16(cs, st);
    for (int r = 0; r < 4; r++) g_full_round(cs, st, rc_first + 16 * r);
    for (int r = 0; r < 14; r++) {
        uint32_t first_only = cs_hadamard(cs, st[0], 1);
        uint32_t k = const_qm31(cs, q_of(0, 1, 1, 1));
        uint32_t without_first = cs_hadamard(cs, st[0], k);
        uint32_t rc = const_m31(cs, rc_partial[r]);
        first_only = cs_add(cs, first_only, rc);
        first_only = g_pow5(cs, first_only);
        st[0] = cs_add(cs, first_only, without_first);
        uint32_t s1 = cs_grandsum(cs, st[0], st[1]);
        uint32_t s2 = cs_grandsum(cs, st[2], st[3]);
        uint32_t sum = cs_add(cs, s1, s2);
        for (int i = 0; i < 4; i++) {
            m31 d[4];
            for (int j = 0; j < 4; j++) { int idx = 4 * i + j; d[j] = idx == 0 ? 3 : (m31)1 << (idx + 1); } /* parameters.rs:6-23 */
            uint32_t kd = const_qm31(cs, q_of(d[0], d[1], d[2], d[3]));
            uint32_t v = cs_hadamard(cs, st[i], kd);
            st[i] = cs_add(cs, sum, v);
        }
    }
    for (int r = 0; r < 4; r++) g_full_round(cs, st, rc_last + 16 * r);
    memcpy(out4, st, sizeof st);
    return RSV_OK;
}

/* ------------------------------------------------------------------------------------------ :410-598 */
/* 0 when every row satisfies its gate equation, else 1 + index of the first row that does not */
size_t rsvo_ecs_check_arithmetics(const rsvo_ecs* cs) {
    for (size_t i = 0; i < cs->n_rows; i++) {
        const row* g = &cs->rows[i];
        q31 a = cs->vars[g->a], b = cs->vars[g->b], c = cs->vars[g->c];
        q31 m4r = q_m4(q_hadamard(a, b)), p4 = q_pow4(a), had = q_hadamard(a, b), gs = q_grand(a, b);
        int ok;
        if (g->op2 == 0 && g->op3 == 0 && g->op4 == 0) /* c = op1 (a + b) + (1 - op1) a b */
            ok = q_eq(c, q_add(q_scale(q_add(a, b), g->op1), q_scale(q_mul(a, b), f_sub(1, g->op1))));
        else if (g->op1 != 1) ok = 0;
        else if (g->op2 == 0 && g->op3 == 0 && g->op4 == 1) ok = q_eq(c, had);
        else if (g->op2 == 1 && g->op3 == 1 && g->op4 == 0) ok = q_eq(b, p4) && q_eq(c, m4r);
        else if (g->op2 == 1 && g->op3 == 0 && g->op4 == 1) ok = q_eq(b, p4) && q_eq(c, had);
        else if (g->op2 == 0 && g->op3 == 1 && g->op4 == 0) ok = q_eq(c, m4r);
        else if (g->op2 == 0 && g->op3 == 1 && g->op4 == 1) ok = q_eq(c, gs);
        else ok = 0;
        if (!ok) return i + 1;
    }
    return 0;
}
