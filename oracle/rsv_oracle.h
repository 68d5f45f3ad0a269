/*
 * rsv_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the recursive-stwo verify path
 * (see rsv_oracle.c for the per-function reference citations).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (recursive-stwo_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED by (a) the reference's literal Poseidon2 known-answer
 * test (primitives/poseidon31/src/implementation.rs:157-172) and (b) the
 * reference's 15 Poseidon-channel proof fixtures, each of which must ACCEPT
 * under the config written in the reference source, plus the SHA-256-channel
 * fixture which must REJECT.  The Rust reference itself cannot
 * be built in this image (no cargo/rustc; stwo dependency un-vendored).
 */
#ifndef RSV_ORACLE_H_
#define RSV_ORACLE_H_
#include "../include/rsv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same semantics as the rsv_* functions of include/rsv.h, computed on the CPU. */
int rsvo_poseidon2_permute(const uint32_t* in16, uint32_t* out16, size_t n);
int rsvo_poseidon2_half_permute(const uint32_t* left8, const uint32_t* right8,
                                const uint8_t* swap, uint32_t* out_rate8,
                                uint32_t* out_cap8, size_t n);
int rsvo_merkle_hash_node(const uint32_t* left8, const uint32_t* right8,
                          const uint32_t* cols, size_t n_cols, uint32_t* out8, size_t n);
int rsvo_merkle_path_root(const uint32_t* query, const uint32_t* sib8, const uint32_t* cols,
                          const uint32_t* n_cols_at, uint32_t depth, uint32_t* out_root8,
                          size_t n);
int rsvo_grind_nonce(const uint8_t* proof, size_t len, uint64_t start, uint64_t max_tries, int want_duplicate_query,
                     uint64_t* nonce);
int rsvo_transcript(const uint8_t* proof, size_t len, uint32_t* out, size_t cap);
int rsvo_verify_batch(const uint8_t* blob, const uint64_t* offsets, size_t n,
                      const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi,
                      uint8_t* accept, uint8_t* reason);

/* Extra probes used by the tests. */
/* Number of Poseidon2 permutations executed by the calling thread since the
 * last reset (transcript + batched Merkle), SURVEY App. C. */
uint64_t rsvo_perm_count(void);
void rsvo_perm_count_reset(void);
/* Field probes: out = a*b, out = a^-1 in QM31 (4 words each). */
void rsvo_qm31_mul(const uint32_t* a, const uint32_t* b, uint32_t* out);
void rsvo_qm31_inv(const uint32_t* a, uint32_t* out);
/* CanonicCoset(log).circle_domain().at(bit_reverse(q, log)) -> (x, y). */
void rsvo_domain_point(uint32_t log_size, uint32_t q, uint32_t* xy);
int rsvo_field_op(int op, const uint32_t* a4, const uint32_t* b4, uint32_t* out4, size_t n);
int rsvo_oods_eval(const uint32_t* samples4, const uint32_t* params26, uint32_t* out8, size_t n);
int rsvo_line_eval(const uint32_t* coeffs4, uint32_t log_n, const uint32_t* x, uint32_t* out4, size_t n);
/* Per-query intermediate values of one proof (for kernel-level parity tests):
 * out receives, for every query j in transcript order, the DEEP-quotient
 * answers for the distinct column log sizes in descending order
 * (n_sizes*4 words) followed by the value entering the last-layer check (4 words).
 * Returns the number of words written or a negative rsv_status. */
int rsvo_query_values(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi,
                      uint32_t* out, size_t cap);

/* per-query authentication paths of the four commitment trees (transcript query order):
 * sib [4][n_queries][M][8] (k-th level above the leaf at index k), pos [4][n_queries], depth4 [4]. */
int rsvo_trace_paths(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* sib,
                     size_t cap, uint32_t* pos, uint32_t* depth4, uint32_t* n_queries);
int rsvo_trace_cols(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* cols, size_t cap,
                    uint32_t* n_queries);
int rsvo_query_dump(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out, size_t cap,
                    uint32_t* n_inner_out, uint32_t* nq_out);
int rsvo_fri_folded(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out, size_t cap,
                    uint32_t* n_sizes, uint32_t* n_queries);

/* SURVEY 8f.1, pair trees (layout: see rsv_oracle.c). */
int rsvo_fri_paths(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* sib,
                   size_t cap, uint32_t* cols, uint32_t* n_trees, uint32_t* n_queries);

/* SURVEY 8f.1, second half: PoseidonFlow of the circuit that verifies `proof` — one record of 33 words
 * (left8 | right8 | out_rate8 | out_cap8 | swap) per Poseidon2HalfVar::permute invocation, in invocation order
 * (layout and order: rsv_oracle.c).  *count receives the number of invocations (also when cap is too small). */
int rsvo_poseidon_flow(const uint8_t* proof, size_t len, const rsv_public_input* pi, size_t n_pi, uint32_t* out,
                       size_t cap, size_t* count);

/* ---- rsv_emulated.c: the emulated Poseidon2 gadget over a minimal Plonk-without-Poseidon constraint system
 * (primitives/poseidon31/src/emulated.rs:80-221, constraint_system/src/plonk_without_poseidon.rs) */
typedef struct rsvo_ecs rsvo_ecs;
enum { RSVO_VAR_FIXED = 0, RSVO_VAR_WITNESS = 1, RSVO_VAR_CONSTANT = 2, RSVO_VAR_GATE = 3 };
const uint32_t* rsvo_round_constants(int which);
rsvo_ecs* rsvo_ecs_new(void);
void rsvo_ecs_free(rsvo_ecs* cs);
size_t rsvo_ecs_n_vars(const rsvo_ecs* cs);
size_t rsvo_ecs_n_rows(const rsvo_ecs* cs);
uint32_t rsvo_ecs_new_witness_m31(rsvo_ecs* cs, uint32_t v);
uint32_t rsvo_ecs_new_witness_qm31(rsvo_ecs* cs, const uint32_t* v4);
uint32_t rsvo_ecs_qm31_from_m31(rsvo_ecs* cs, const uint32_t* m31_vars4);
int rsvo_ecs_permute_emulated(rsvo_ecs* cs, const uint32_t* left2, const uint32_t* right2, int swap_mode,
                              uint32_t bit_var, uint32_t* out4);
size_t rsvo_ecs_check_arithmetics(const rsvo_ecs* cs);
/* vars4: [n_vars][4]; kind: [n_vars] RSVO_VAR_*; rows7: [n_rows][7] = a, b, c wires, op1..op4 */
void rsvo_ecs_export(const rsvo_ecs* cs, uint32_t* vars4, uint8_t* kind, uint32_t* rows7);
int rsvo_ecs_set_vars(rsvo_ecs* cs, size_t first, const uint32_t* vars4, size_t n);

#ifdef __cplusplus
}
#endif
#endif
