/*
 * rsv.h — C-ABI drop-in boundary of the MI355X-native batch verifier for
 * Poseidon31-channel "Plonk-with-Poseidon" Circle-STARK proofs.
 *
 * The reference (Bitcoin-Wildlife-Sanctuary/recursive-stwo) has no FFI of its
 * own: the verify path is reached through Rust generics.  Every entry point
 * below names the reference item whose *values* it replaces (paths relative to
 * the reference root); INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - All field elements are canonical M31 words (0 <= w < 2^31-1) in
 *     little-endian u32.  QM31 = 4 words (a0,a1,b0,b1) for (a0+a1*i)+(b0+b1*i)*u.
 *     A hash is 8 words.
 *   - Return value: RSV_OK (0) on success, negative rsv_status on API misuse
 *     or device failure.  A proof that does not verify is DATA
 *     (accept[i]=0, reason[i]=why), never an error and never an abort.
 *   - The caller owns every buffer.  Functions taking `device` copy host
 *     buffers to that HIP device and back (stream-ordered, synchronous on
 *     return).  `_dev` variants take device pointers already resident in HBM
 *     and enqueue on the context's stream.
 *   - There is NO CPU fallback in this library: if no HIP device is usable the
 *     call fails with RSV_E_DEVICE.
 *   - Thread-safety: distinct rsv_ctx objects may be used concurrently; one
 *     ctx is single-threaded (mirrors the `!Send` reference types,
 *     constraint_system/src/lib.rs:33).
 */
#ifndef RSV_H_
#define RSV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSV_M31_P 0x7fffffffu
#define RSV_ABI_VERSION 6

typedef enum rsv_status {
    RSV_OK = 0,
    RSV_E_NULL = -1,     /* required pointer is NULL */
    RSV_E_SIZE = -2,     /* inconsistent sizes / offsets not monotone / n too large */
    RSV_E_DEVICE = -3,   /* no such HIP device, or a HIP call failed */
    RSV_E_CAP = -4,      /* output capacity too small */
    RSV_E_RANGE = -5,    /* input word not a canonical M31 */
    RSV_E_UNAVAILABLE = -6, /* an optional run-time dependency (RCCL, for rsv_exchange_*) could not be loaded */
    RSV_E_NOMEM = -7     /* host memory for the call's staging could not be allocated (no verdict was written) */
} rsv_status;

/* Why proof i was rejected.  Order = the order in which the reference's
 * stages would panic (examples/single-proof/src/main.rs:33-82).
 *
 * Reason codes are this library's extension: the reference panics at the first failed check, so only accept / reject
 * is pinned by its fixtures; which code a rejected proof gets is defined here and checked against this repo's oracle.
 *
 * KNOWN DIVERGENCE from "bit-exact accept / reject on identical bytes" (deliberate, on the safe side): every
 * field-element word of a proof must be canonical (< 2^31-1), otherwise the proof is RSV_R_PARSE.  bincode
 * deserialises M31(u32) unchecked and stwo's release-mode arithmetic reduces such a word (partial_reduce maps P to
 * 0), so the Rust verifier most likely ACCEPTS a genuine proof in which a 0 word has been re-encoded as P
 * (0x7fffffff), and panics on overflow in debug builds for larger words; this library rejects all of them.  No
 * reference fixture contains such a word, and without a Rust toolchain the reference's behaviour on one cannot be
 * observed here.  Exempt (not field elements, read by nothing else): the two halves of the 64-bit proof-of-work
 * nonce and the proof's final word, last_layer_poly.log_size. */
typedef enum rsv_reason {
    RSV_R_OK = 0,
    RSV_R_PARSE = 1,        /* bincode shape / config mismatch (examples/single-proof/src/main.rs:24-31), a non-canonical field
                               element, or a header beyond this library's shape limits (RSV_MAX_*, below) */
    RSV_R_POW = 2,          /* components/recursive/fiat_shamir/src/lib.rs:115-117 */
    RSV_R_LOGUP = 3,        /* components/recursive/fiat_shamir/src/lib.rs:133-141 */
    RSV_R_COMPOSITION = 4,  /* components/recursive/composition/src/lib.rs:106-120 */
    RSV_R_DUP_QUERY = 5,    /* components/recursive/answer/src/lib.rs:190-195 */
    RSV_R_MERKLE_T0 = 6,    /* components/recursive/answer/src/lib.rs:214-258 (tree 0..3) */
    RSV_R_MERKLE_T1 = 7,
    RSV_R_MERKLE_T2 = 8,
    RSV_R_MERKLE_T3 = 9,
    RSV_R_FRI_FIRST = 10,   /* components/recursive/folding/src/lib.rs:23-54 */
    RSV_R_FRI_INNER = 11,   /* components/recursive/folding/src/lib.rs:135-192 */
    RSV_R_FRI_LAST = 12     /* components/recursive/folding/src/lib.rs:194-204 */
} rsv_reason;

/* stwo PcsConfig{pow_bits, FriConfig::new(log_last_layer_degree_bound,
 * log_blowup_factor, n_queries)} as written in
 * examples/multi-proofs/src/main.rs:173-196. */
typedef struct rsv_pcs_config {
    uint32_t pow_bits;
    uint32_t log_blowup_factor;
    uint32_t log_last_layer_degree_bound;
    uint32_t n_queries;
} rsv_pcs_config;

/* SHAPE LIMITS of this library (the reference has none: its vectors grow with the proof).  Tables in HBM and LDS are
 * sized by them.
 *   A CONFIGURATION beyond them is refused: every entry point that takes an rsv_cfg_set returns RSV_E_SIZE (and writes
 *   no verdict) unless every configuration of the set has
 *       1 <= n_queries <= RSV_MAX_QUERIES,  1 <= log_blowup_factor <= RSV_MAX_LOG_BLOWUP,
 *       log_last_layer_degree_bound <= RSV_MAX_LOG_LAST_LAYER,  pow_bits <= RSV_MAX_POW_BITS;
 *   rsv_cfg_check tells without a device.
 *   A PROOF whose own header goes beyond them — component log sizes (log_size_plonk, log_size_poseidon) outside
 *   1 .. RSV_MAX_COMPONENT_LOG, or a largest column log size M = max(log_size_plonk + 1, log_size_poseidon + 2) +
 *   log_blowup_factor above RSV_MAX_LOG_SIZE (which also bounds the FRI inner layers by RSV_MAX_FRI_INNER) — is data, not
 *   misuse: it is rejected with RSV_R_PARSE like any proof this library cannot read, although the reference would verify
 *   it if it is genuine.  (Every fixture of the reference: n_queries 8 .. 80, M 21 .. 28, 7 .. 12 inner layers.)
 *   The witness entry points (rsv_witness_*) additionally need log_size_plonk, log_size_poseidon <= RSV_MAX_WITNESS_LOG. */
#define RSV_MAX_QUERIES 128
#define RSV_MAX_LOG_BLOWUP 16
#define RSV_MAX_LOG_LAST_LAYER 16
#define RSV_MAX_POW_BITS 30
#define RSV_MAX_COMPONENT_LOG 28
#define RSV_MAX_LOG_SIZE 30
#define RSV_MAX_FRI_INNER 28
#define RSV_MAX_WITNESS_LOG 24
/* RSV_OK if the library can verify proofs under *cfg, RSV_E_SIZE if it is beyond the limits above, RSV_E_NULL. */
int rsv_cfg_check(const rsv_pcs_config* cfg);

/* The configuration(s) the caller expects — REQUIRED by every entry point that produces a verdict.  The reference
 * never takes the configuration from the proof: FiatShamirHints::new(&proof, config, ..)
 * (components/hints/src/fiat_shamir.rs:69-74), examples/multi-proofs/src/main.rs:173-196 pass it in; a verifier that
 * trusted the serialized words would let a forger choose pow_bits = 0, n_queries = 1.  Proof i must carry exactly
 * cfgs[cfg_of ? cfg_of[i] : 0] in its header, else it is rejected with RSV_R_PARSE.
 *   cfgs    n_cfgs (1..RSV_MAX_CFGS) configurations, HOST memory
 *   cfg_of  one index per proof (a mixed batch, e.g. the 6 configurations of the multi-proofs chain) or NULL;
 *           same residency as the blob: device memory for the _dev entry points, host memory otherwise.
 *           An index >= n_cfgs rejects the proof (RSV_R_PARSE). */
#define RSV_MAX_CFGS 16
typedef struct rsv_cfg_set {
    const rsv_pcs_config* cfgs;
    uint32_t n_cfgs;
    const uint8_t* cfg_of;
} rsv_cfg_set;

/* One public input `(wire index, QM31 value)` as passed to
 * FiatShamirResults::compute(.., inputs) (components/recursive/fiat_shamir/src/lib.rs:31-36). */
typedef struct rsv_public_input {
    uint32_t idx;
    uint32_t value[4];
} rsv_public_input;

/* Opaque per-device context: HIP stream + reusable HBM workspace. */
typedef struct rsv_ctx rsv_ctx;

int rsv_abi_version(void);
/* Number of usable HIP devices (0 if none; never fails). */
int rsv_device_count(void);
int rsv_ctx_create(int device, rsv_ctx** out);
void rsv_ctx_destroy(rsv_ctx* ctx);
/* Block until everything enqueued on the context's stream has finished. */
int rsv_ctx_synchronize(rsv_ctx* ctx);
/* The context's hipStream_t (as void*), so a caller can order its own work. */
void* rsv_ctx_stream(rsv_ctx* ctx);
/* Stream ordering with the caller's own HIP work (the context enqueues on private non-blocking streams):
 *   rsv_ctx_wait_stream   everything enqueued so far on `hip_stream` (a hipStream_t; NULL = the legacy default stream)
 *                         happens before whatever is enqueued on the context next — call it after producing the
 *                         blob / offsets / output buffers with your own kernels or copies, before a _dev entry point;
 *   rsv_stream_wait_ctx   the reverse: `hip_stream` waits for everything the context has enqueued so far.
 * Neither blocks the host. */
int rsv_ctx_wait_stream(rsv_ctx* ctx, void* hip_stream);
int rsv_stream_wait_ctx(rsv_ctx* ctx, void* hip_stream);

/* Tuning / diagnostic knobs.  The library never reads the environment: a knob changes only through this call, on one
 * context, or — ctx == NULL — as the process default that contexts created LATER inherit (the host-pointer
 * convenience entry points create a context of their own, so that is how the tests steer them).  Every knob only
 * selects between kernel forms / launch layouts that give bit-identical verdicts (the parity tests run each of them
 * against the oracle); 0 is always "automatic" (= what production runs).  Returns RSV_E_SIZE for an unknown option,
 * RSV_E_RANGE for a value outside the listed range. */
typedef enum rsv_option {
    RSV_OPT_TRANSCRIPT_FORM = 1,  /* 0 auto (by batch size), 1 one proof per 16-lane DPP row, 2 one proof per lane */
    RSV_OPT_TRANSCRIPT_SPLIT = 2, /* 0 auto, 1 one launch, 2 front half beside the parser + back half (row form only) */
    RSV_OPT_OODS_FORM = 3,        /* 0 auto, 1 row, 2 lane */
    RSV_OPT_QCONST_FORM = 4,      /* 0 auto (<= 4 096 proofs: four 16-lane rows per proof, <= 24 576: one row, else lane), 1 one row per proof, 2 lane */
    RSV_OPT_PLAN_FORM = 5,        /* 0 / 1 one lane per (proof, query), 2 one lane per proof */
    RSV_OPT_TREE_CAP = 6,         /* 0 / 1 dense top-of-tree cap, 2 every lane walks its path to the root */
    RSV_OPT_OVERLAP_TREES = 7,    /* 0 auto, 1 FRI trees beside the trace trees (single-group batches), 2 behind them */
    RSV_OPT_WS_BUDGET_MB = 8,     /* 1 .. 2^20: budget of the per-query workspace (default 8192); larger batches are cut into groups */
    RSV_OPT_PERM_WG_PER_CU = 9,   /* 1 .. 32: workgroups per CU of the grid-stride rsv_poseidon2_permute kernel (default 24) */
    RSV_OPT_HOST_CHUNK_MB = 10,   /* 1 .. 16384: staging chunk of rsv_verify_batch_host (default 256) */
    RSV_OPT_HOST_THREADS = 11,    /* 0 = min(cores, 4), else 1 .. 64 gather threads of rsv_verify_batch_host */
    RSV_OPT_DEBUG_LOG = 12,       /* 0 / 1: print failing HIP calls to stderr (process-wide, ctx ignored) */
    RSV_OPT_CRITICAL_CHAIN = 13,  /* 0 auto, 1 the step's chain of dependent kernels on one stream, 2 the two-stream layout */
    RSV_OPT_DEVICE_ORDER = 14,    /* 0 / 1 batches under one configuration: slot order by shape on the device, no host round
                                     trip inside the call; 2 the host-side bucketing of multi-configuration batches */
    RSV_OPT_GRAPH = 15,           /* 0 / 2 off; 1 (experiment) a call repeated with identical arguments — same buffers, sizes,
                                     configuration, public inputs — is captured into a HIP graph on its second sighting
                                     and replayed afterwards; rsv_last_stage_times then reports the last plain call */
    RSV_OPT_WITNESS_LAYOUT = 16,  /* rsv_witness_eval_dev's d_variables: 0 / 1 [proof][variable] (the reference's vector per
                                     proof); 2 [variable][proof] — what the level kernels write: no transpose (a third of
                                     the traffic) and no second copy in scratch, for consumers that gather for many proofs */
    RSV_OPT_WITNESS_SMALL_MAX = 17, /* rsv_witness_eval_dev: 0 default (by batch and program size), else 1 + the largest batch whose
                                     program runs in ONE launch (a workgroup per few proofs walks all levels) instead of one
                                     launch per level; 1 = never */
    RSV_OPT_WITNESS_SMALL_LOG = 18, /* 0 default, else 1 + log2(proofs per workgroup) of that form, 1 .. 7 */
    RSV_OPT_WITNESS_WALK_LOG = 20, /* rsv_witness_eval_dev, batches that run the program level by level: the levels behind the
                                     program's wide head in ONE launch, a workgroup per 2^k proofs: 0 auto, 1 off, else 1 + k (k = 1 .. 6) */
    RSV_OPT_FLOW_CAP = 21,        /* passes that emit the PoseidonFlow: 0 / 1 a node several queries share is hashed once and the record
                                     of every query through it written from there, 2 every lane hashes its whole path itself */
    RSV_OPT_PAIR_ORDER = 22,      /* 0 / 1 a k_pair_merkle launch that is resident all at once deals its FRI trees (grid rows) out over the
                                     compute units by depth (the dispatcher does not balance such a launch), 2 grid row y = tree y */
    RSV_OPT_TREE_PACE = 23,       /* 0 auto (small batches: 2, a few hundred proofs and fewer: 3), 1 the Merkle kernels call the permutation
                                     instance with wait states behind its multiplies (pays with several waves per SIMD), 2 the one
                                     without (a wave nearly alone), 3 the row form: 16 threads per Merkle path share every permutation
                                     (a launch of a few waves: the walk is a chain of dependent permutations, the row form's a
                                     quarter as long); not with more than 32 queries (then: 2, or 1 when the flow is written) */
    RSV_OPT_QUERY_FORM = 25,      /* 0 auto (by batch size), 1 the quotient / fold kernel with a row of 16 threads per query that split its sums
                                     (chain layout only; not with more than 32 queries), 2 with one lane per query */
    RSV_OPT_STAGE_TIMES = 24,     /* 0 / 2 off; 1 record a HIP event pair around every stage of a verify call, which is what
                                     rsv_last_stage_times reads.  Off by default: the records cost ~8 us of queue time per stage */
    RSV_OPT_CAP_MID = 26,         /* with the cap's top in kernels of its own (RSV_OPT_CAP_TOP): 0 auto — a bucket of proofs whose dense cap levels
                                     fill the tree kernels' waves badly (80, 27, 11, 10 queries) hands its nodes over at the cap level, a
                                     lane per subtree walks the middle levels (k_cap_mid), k_cap_top the rest; 1 every bucket does, 2 none */
    RSV_OPT_PERM_FORM = 27,       /* rsv_poseidon2_permute_dev, experiments: 0 production (the out-of-line instance the verify kernels call), 1 the same
                                     inlined, 2 inlined without wait states (recursive-stwo_amd/csrc/primitives.hpp: k_permute) */
    RSV_OPT_OODS_EARLY = 28,      /* chain layout, experiments: 0 / 2 the OODS check behind the trace trees on the side stream; 1 on a third
                                     stream right behind the transcript (measured: slower from 1 024 to 4 096 proofs, not taken) */
    RSV_OPT_TREE_ORDER = 29,      /* workgroup order of the lane-form Merkle kernels: 0 / 2 tree by tree (grid row y = tree); 1 (measured, slower:
                                     less HBM traffic, more time) the trees of a workgroup of proofs side by side and on one XCD, so that
                                     their plan tables are fetched once per L2, not once per tree */
    RSV_OPT_CAP_TOP = 19          /* 0 auto (batches of >= 1 024 proofs), 1 the last two or three levels of every Merkle tree in a
                                     kernel of their own (one lane per tree), 2 inside the tree kernels (dense top-of-tree cap) */
} rsv_option;
int rsv_ctx_set_option(rsv_ctx* ctx, int option, long long value);

/* ---- a3: Poseidon2-M31 width-16 permutation -------------------------------
 * Replaces poseidon2_permute (primitives/poseidon31/src/implementation.rs:108-149).
 * n states of 16 words, state-major (state i at in16 + 16*i). */
int rsv_poseidon2_permute(const uint32_t* in16, uint32_t* out16, size_t n, int device);
/* Device-resident form: enqueued on the context's stream, no host round trip.  Inputs >= P are still permuted
 * (as canonical-mod-P garbage); if d_bad (device u32, may be NULL) is given it is set to 1 when any input word
 * was not canonical and left untouched otherwise — the caller zeroes it and reads it when it synchronises. */
int rsv_poseidon2_permute_dev(rsv_ctx* ctx, const uint32_t* d_in16, uint32_t* d_out16, size_t n, uint32_t* d_bad);

/* ---- a4: Poseidon2HalfVar::permute with swap / rate / capacity ------------
 * Replaces Poseidon2HalfVar::permute(left,right,_,_,is_swap)
 * (primitives/poseidon31/src/lib.rs:282-423), values only.
 * For each i: state = swap[i] ? right_i||left_i : left_i||right_i; permute;
 * out_rate8 = state[0..8], out_cap8 = state[8..16].  swap, out_rate8 or
 * out_cap8 may be NULL (no swap / result ignored). */
int rsv_poseidon2_half_permute(const uint32_t* left8, const uint32_t* right8,
                               const uint8_t* swap, uint32_t* out_rate8,
                               uint32_t* out_cap8, size_t n, int device);

/* ---- f4: gate values of the emulated Poseidon2 ------------------------------
 * Replaces the value side of poseidon_permute_emulated(left, right, is_swap)
 * (primitives/poseidon31/src/emulated.rs:80-221): the permutation written as M4 / pow5m4 / pow5 / Hadamard /
 * grand-sum / add / mul gates of the Plonk-without-Poseidon constraint system
 * (constraint_system/src/plonk_without_poseidon.rs:113-305).  For each of n permutations, rows receives the QM31
 * value of every variable the gadget appends to the circuit's witness, in allocation order, for a circuit whose
 * constants are already cached (primitives/fields/src/qm31.rs:39-73: every call but a circuit's first):
 *   rows[p][0 .. 12)    the swap gates, present when is_swap = Some(..) and zero when it is None
 *   rows[p][12 .. 413)  the 401 variables of the permutation itself; rows[p][409 .. 413) are the output state
 *                       (left half in the first two)
 *   rows[p][413 .. 416) zero: padding to RSV_EMU_STRIDE rows, so that each permutation's rows are 52 whole 128-byte
 *                       lines (the kernel is bound by HBM writes and stores whole aligned lines)
 * left8 / right8: n x 8 canonical words (a half = two QM31 of four words each).  swap: NULL (all None) or n bytes,
 * 0 = None, 1 = Some((false, _)), 2 = Some((true, _)); other values or words >= P give RSV_E_RANGE. */
#define RSV_EMU_SWAP_ROWS 12
#define RSV_EMU_ROWS 413
#define RSV_EMU_STRIDE 416
int rsv_poseidon2_emulated(const uint32_t* left8, const uint32_t* right8, const uint8_t* swap,
                           uint32_t* rows /* [n][RSV_EMU_STRIDE][4] */, size_t n, int device);
/* Device-resident form on the context's stream.  d_left8 / d_right8 / d_rows must be 16-byte aligned (d_rows on a
 * 128-byte line for full speed).  d_bad
 * (device u32, may be NULL) is set to 1 on a range error and left untouched otherwise, as for
 * rsv_poseidon2_permute_dev. */
int rsv_poseidon2_emulated_dev(rsv_ctx* ctx, const uint32_t* d_left8, const uint32_t* d_right8, const uint8_t* d_swap,
                               uint32_t* d_rows, size_t n, uint32_t* d_bad);

/* ---- a5: Poseidon31 Merkle hasher -----------------------------------------
 * Replaces Poseidon31MerkleHasherVar::{hash_tree, hash_tree_with_column,
 * hash_m31_columns_get_rate, ...} (primitives/merkle/src/lib.rs:9-181) ==
 * stwo Poseidon31MerkleHasher::hash_node(children, columns).
 * n nodes; node i has children left8+8*i / right8+8*i (both NULL => leaves)
 * and n_cols column words at cols + n_cols*i (n_cols may be 0 iff children given). */
int rsv_merkle_hash_node(const uint32_t* left8, const uint32_t* right8,
                         const uint32_t* cols, size_t n_cols, uint32_t* out8,
                         size_t n, int device);

/* ---- a1 / a2 / a8 / last layer of a12: arithmetic probes ------------------------
 * The device functions the verify kernels are built from, exposed one lane per item so that a binding (and the
 * parity tests) can check them in isolation.  Elements are QM31 = 4 words (a0, a1, b0, b1) = (a0 + a1 i) +
 * (b0 + b1 i) u, all canonical (else RSV_E_RANGE).
 *   RSV_F_QADD/QSUB/QMUL  out = a (+,-,*) b                    primitives/fields/src/qm31.rs:87-249
 *   RSV_F_QINV            out = 1/a                            qm31.rs:360-366
 *   RSV_F_MMUL / MINV     first word only: a0*b0, 1/a0         primitives/fields/src/m31.rs:62-115,140-156
 *   RSV_F_CMUL / CINV     first two words: CM31 product, 1/a   primitives/fields/src/cm31.rs:87-192
 *   RSV_F_QMULI / QMULU   a*i, a*u                             qm31.rs:402-418
 *   RSV_F_QPOW            a ^ (first word of b, any u32)       qm31.rs (QM31Var::pow; test :489)
 * b4 may be NULL for the unary operations. */
enum { RSV_F_QADD = 0, RSV_F_QSUB, RSV_F_QMUL, RSV_F_QINV, RSV_F_MMUL, RSV_F_MINV, RSV_F_CMUL, RSV_F_CINV,
       RSV_F_QMULI, RSV_F_QMULU, RSV_F_QPOW };
int rsv_field_op(int op, const uint32_t* a4, const uint32_t* b4, uint32_t* out4, size_t n, int device);
/* a8: xy[2i], xy[2i+1] = CanonicCoset(log_size).circle_domain().at(bit_reverse(q[i], log_size)) — the point
 * PointCarryingQueryVar carries for position q[i] (primitives/query/src/lib.rs:57-168, circle/src/lib.rs:44-131;
 * reference test circle/src/lib.rs:264).  q[i] is masked to log_size bits; 1 <= log_size <= 30. */
int rsv_domain_points(uint32_t log_size, const uint32_t* q, uint32_t* xy, size_t n, int device);
/* a10 on its own: the OODS composition evaluation of CompositionCheck::compute
 * (components/recursive/composition/src/lib.rs:60-120, plonk.rs:8-82, poseidon.rs:73-241) for n items.
 *   samples4  [n][142][4]  the sampled values, flattened tree-major / column-major / sample-minor
 *   params26  [n][26]      log_size_plonk, log_size_poseidon (1..28), plonk_total_sum, poseidon_total_sum, z, alpha,
 *                          random_coeff, oods_point.x                          (QM31 = 4 words each)
 *   out8      [n][8]       the accumulator over the 86 constraints | left + right * pi^(bound-2)(x): the proof passes
 *                          the composition check iff the two are equal */
int rsv_oods_eval(const uint32_t* samples4, const uint32_t* params26, uint32_t* out8, size_t n, int device);
/* The last-layer comparison of FoldingResults::compute (components/recursive/folding/src/lib.rs:194-204) on its own:
 * ok[i] = 1 iff the polynomial (2^log_n QM31 coefficients) evaluated at x[i] equals folded4[i] — the device function
 * whose failure is RSV_R_FRI_LAST, a reason no mutated proof reaches (the polynomial is hashed before the proof of
 * work, so every mutation stops earlier). */
int rsv_last_layer_check(const uint32_t* coeffs4, uint32_t log_n, const uint32_t* x, const uint32_t* folded4, uint8_t* ok,
                         size_t n, int device);
/* LinePolyVar::eval_at_point (primitives/line/src/lib.rs:39-67; reference test :82): one polynomial of 2^log_n
 * QM31 coefficients (log_n <= 16) evaluated at n points x[i] (M31). */
int rsv_line_eval(const uint32_t* coeffs4, uint32_t log_n, const uint32_t* x, uint32_t* out4, size_t n, int device);

/* ---- a9: one authentication path per query --------------------------------
 * Replaces SinglePathMerkleProofVar::verify
 * (components/recursive/data_structures/src/lib.rs:315-354).
 * n independent paths of equal `depth`; path i: query position query[i],
 * siblings sib8 + 8*depth*i (leaf level first), column words laid out per
 * level: level h (depth..0) has n_cols_at[h] words for every path, packed
 * level-major from the leaf level down (cols + i*sum(n_cols_at)).
 * out_root8 receives the recomputed root (8 words per path). */
int rsv_merkle_path_root(const uint32_t* query, const uint32_t* sib8,
                         const uint32_t* cols, const uint32_t* n_cols_at /* depth+1 */,
                         uint32_t depth, uint32_t* out_root8, size_t n, int device);

/* ---- a6/a7: Fiat-Shamir transcript (parity probe) -------------------------
 * Replaces FiatShamirResults::compute (components/recursive/fiat_shamir/src/lib.rs:31-176).
 * out layout (u32 words):
 *   [0]      reason after transcript stage (RSV_R_OK / PARSE / POW)
 *   [1]      n_fri_alphas = 1 + n_inner_layers
 *   [2]      n_queries
 *   [3]      M = log size of the largest committed column (query bits)
 *   [4..8)   z          [8..12)  alpha       [12..16) random_coeff
 *   [16..20) oods_t     [20..24) oods.x      [24..28) oods.y
 *   [28..32) after_sampled_values_random_coeff
 *   [32..40) channel digest after the proof-of-work mix
 *   [40..40+4*n_fri_alphas)  fri alphas
 *   then n_queries raw query words (before masking to M bits).
 * Returns RSV_E_CAP if cap (in words) is too small.
 * PROBE, not a verdict: it replays the transcript under the configuration words serialized in the proof (no
 * rsv_cfg_set), so word [0] == RSV_R_OK says nothing about the proof's security level. */
int rsv_transcript(const uint8_t* proof, size_t len, uint32_t* out, size_t cap, int device);

/* ---- full verify: a1-a13 --------------------------------------------------
 * Replaces the stage sequence FiatShamirResults::compute →
 * CompositionCheck::compute → AnswerResults::compute → FoldingResults::compute
 * (examples/single-proof/src/main.rs:48-82) on a batch of serialized
 * PlonkWithPoseidonProof<Poseidon31MerkleHasher> (bincode).
 *   blob     concatenated proof bytes; proof i = blob[offsets[i] .. offsets[i+1])
 *   cfg      REQUIRED (RSV_E_NULL otherwise): the configuration(s) the caller expects, see rsv_cfg_set; a proof
 *            whose embedded configuration differs is rejected with RSV_R_PARSE
 *   pi,n_pi  public inputs shared by the whole batch
 *   accept   n bytes, 1 = verified        reason  n bytes of rsv_reason (may be NULL)
 */
int rsv_verify_batch(const uint8_t* blob, const uint64_t* offsets, size_t n,
                     const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi,
                     uint8_t* accept, uint8_t* reason, int device);

/* Same, inputs and outputs resident in HBM (d_ = device pointers); enqueued on ctx's streams.  d_offsets must be 8-byte
 * aligned, d_blob 4-byte aligned and every offset a multiple of 4 (bincode proofs of this type always have
 * 4-byte-multiple lengths).
 * Host synchronisation: a batch under ONE configuration (cfg->n_cfgs == 1, cfg_of NULL) is enqueued completely and the
 * call returns without waiting for the device (after the first call has grown the workspaces).  A batch under several
 * configurations is bucketed by n_queries on the host: the call waits for the parser (a few hundred microseconds),
 * reads 8 bytes per proof back, and enqueues the rest; so do the calls with per-query path outputs
 * (rsv_trace_paths_dev, rsv_fri_paths_dev, the path / query-value outputs of rsv_verify_hints_dev), whose declared
 * shape is checked on the host.  Either way the verdicts are complete only after rsv_ctx_synchronize /
 * rsv_stream_wait_ctx. */
int rsv_verify_batch_dev(rsv_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets,
                         size_t n, const rsv_cfg_set* cfg, const rsv_public_input* pi,
                         size_t n_pi, uint8_t* d_accept, uint8_t* d_reason);

/* Proofs that start in HOST memory, as the reference's callers hold them: one serialized buffer per proof
 * (bincode::serialize(&proof) -> Vec<u8>, examples/multi-proofs/src/main.rs:69-139).  Chunks of about
 * RSV_OPT_HOST_CHUNK_MB (default 256) MB are gathered into pinned staging memory by worker threads
 * (RSV_OPT_HOST_THREADS, default min(cores, 4)), uploaded by the DMA engine and verified, the three stages overlapping
 * (a job below eight such chunks is cut into eight, none under 32 MB, so that the pipeline has something to overlap);
 * accept / reason are host arrays of n bytes.  Blocks until every verdict is written.  A buffer whose length is not a
 * multiple of 4 (every proof of this type is a whole number of 32-bit words) or exceeds 32 MB (a well-formed proof is
 * below 8 MB) is not uploaded and gets RSV_R_PARSE, like any other malformed proof. */
int rsv_verify_batch_host(rsv_ctx* ctx, const uint8_t* const* proofs, const uint64_t* lens, size_t n,
                          const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, uint8_t* accept,
                          uint8_t* reason);
/* Pinned host memory for proofs.  A caller that can cooperate reads / deserialises its proofs straight into such an
 * arena, back to back in job order: rsv_verify_batch_host (and rsv_multi_verify_batch_host) then recognise every chunk
 * whose proofs lie contiguously inside one arena and let the DMA engine upload it from where it is — no gather copy, no
 * staging ring (the pinned ring is not even allocated when the whole job qualifies).  Anything else (pageable memory,
 * gaps, proofs out of order) goes through the gather as before; the two mix freely chunk by chunk.  hipHostMalloc is
 * slow (about 1 GB/s): allocate once, reuse.  rsv_host_free ignores pointers this library did not hand out. */
int rsv_host_alloc(size_t bytes, void** out);
void rsv_host_free(void* p);

/* ---- SURVEY 8f.1 (next row): per-query authentication paths -----------------
 * Emits, while verifying, the per-query Merkle paths of the four commitment trees in TRANSCRIPT query order —
 * the data SinglePathMerkleProof::from_stwo_proof (components/hints/src/decommit.rs:44-183) derives on the host
 * by re-hashing stwo's batched decommitment.  All n proofs must share one shape: n_queries (>= 4) and
 * max_log = M (log size of the largest column) as declared by the caller, else RSV_E_SIZE.
 *   d_sib  [n][4][n_queries][max_log][8]  sibling hash at the k-th level above the leaf of tree t at index k
 *                                         (entries beyond the tree's depth are not written)
 *   d_pos  [n][4][n_queries]              position of the query at the tree's leaf level
 * Tree depths: max(lp, lq) + log_blowup for trees 0..2, M for tree 3.  accept/reason as in rsv_verify_batch_dev. */
int rsv_trace_paths_dev(rsv_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                        const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, uint32_t n_queries, uint32_t max_log,
                        uint32_t* d_sib, uint32_t* d_pos, uint8_t* d_accept, uint8_t* d_reason);
/* Same on host buffers (copied to `device` and back). */
int rsv_trace_paths(const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg,
                    const rsv_public_input* pi, size_t n_pi,
                    uint32_t n_queries, uint32_t max_log, uint32_t* sib, uint32_t* pos, uint8_t* accept,
                    uint8_t* reason, int device);

/* Same for the FRI trees: the per-query pair paths SinglePairMerkleProof::from_stwo_proof
 * (components/hints/src/folding.rs:93-287) derives on the host.  Tree s = 0 is the first layer (leaf level
 * M = max_log, one QM31 column at each distinct column log size), s = 1 + i inner layer i (leaf level M - 1 - i).
 * The batch must share (n_queries >= 4, max_log, n_inner), else RSV_E_SIZE.
 *   d_sib  [n][1 + n_inner][n_queries][max_log][8]  sibling_hashes[k], k = 0..depth-2: at level depth-1-k the
 *                                                   sibling's hash, or — where that level carries a column —
 *                                                   the hash of the sibling's children
 *   d_cols [n][1 + n_inner][n_queries][3][8]        c-th column level from the top: self value | sibling value */
int rsv_fri_paths_dev(rsv_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                      const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, uint32_t n_queries, uint32_t max_log, uint32_t n_inner,
                      uint32_t* d_sib, uint32_t* d_cols, uint8_t* d_accept, uint8_t* d_reason);
int rsv_fri_paths(const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg,
                  const rsv_public_input* pi, size_t n_pi,
                  uint32_t n_queries, uint32_t max_log, uint32_t n_inner, uint32_t* sib, uint32_t* cols, uint8_t* accept,
                  uint8_t* reason, int device);

/* ---- everything the reference's hint structs need, from ONE verifying pass ------------------------
 * rsv_verify_hints_dev = rsv_verify_batch_dev that also writes whichever of these outputs are non-NULL:
 *   d_transcript [n][RSV_TRANSCRIPT_WORDS]  per proof, what FiatShamirHints (components/hints/src/fiat_shamir.rs:69-256)
 *                                           carries: words [0..40) as rsv_transcript's; [40..40+4*29) the FRI alphas
 *                                           (first layer, then inner layers; zero padded); [156..284) the raw query
 *                                           words in transcript order (zero padded).  Mixed shapes allowed.
 *   d_trace_sib / d_trace_pos               as rsv_trace_paths_dev   (both or neither)
 *   d_trace_cols [n][4][n_queries][64]      optional (with or without d_trace_sib): SinglePathMerkleProof::columns — the query's
 *                                           column values at the leaf level, then those at the lower column log size
 *                                           (the packing rsv_merkle_path_root takes)
 *   d_fri_sib / d_fri_cols                  as rsv_fri_paths_dev     (both, or d_fri_cols alone)
 *   d_fri_folded [n][3][n_queries][4]       optional, with d_fri_sib: FirstLayerHints::folded_evals_by_column
 *                                           (components/hints/src/folding.rs:291-293) — the circle-to-line fold of
 *                                           each query's first-layer pair at the c-th column log size (descending)
 *   d_query_values [n][n_queries][4*(8+n_inner)]  optional parity probe of rows a11 / a12 (values, not only verdicts),
 *                                           per query in transcript order, QM31 each: the DEEP-quotient answers at
 *                                           the (up to 3) column log sizes, descending (answer/src/lib.rs:294-315);
 *                                           their circle-to-line folds (folding/src/lib.rs:57-90); the value entering
 *                                           inner layer i, i < n_inner (:135-144); the value entering the last-layer
 *                                           check and the last-layer polynomial evaluated at the query's point (:194-204)
 *                                           (the library zeroes this buffer first: absent groups and the rows of rejected
 *                                           proofs are zero)
 * Path outputs need a uniform batch of the declared shape (n_queries >= 4, max_log, n_inner), else RSV_E_SIZE.
 *
 * SURVEY 8f.1, second half — the value side of the recursion circuit's Poseidon accelerator:
 *   d_flow       [n][flow_stride][32]  PoseidonFlow (constraint_system/src/plonk_with_poseidon.rs:36,117-128,468-519) of
 *                                      the circuit that verifies proof i: one record per Poseidon2HalfVar::permute
 *                                      invocation (primitives/poseidon31/src/lib.rs:282-423) = the hash words of the four
 *                                      PoseidonEntry: left8 | right8 (the inputs as given; the accelerator applies
 *                                      the swap) | out_rate8 | out_cap8 — one 128-byte line per record; 16-byte aligned
 *   d_flow_swap  [n][flow_stride]      SwapOption::swap of the record (bytes; both or neither with d_flow)
 *   d_flow_count [n]                   optional: records of proof i = rsv_poseidon_flow_count of its shape; 0 when the
 *                                      parser rejected it or flow_stride is too small for it (nothing is written then)
 *   flow_stride                        records the caller allocated per proof
 * Record order = the circuit's invocation order (examples/multi-proofs/src/main.rs:69-139): every channel operation of
 * FiatShamirResults::compute (including the ceil(n_queries / 4) query draws the circuit makes where ceil(n_queries / 8)
 * hold every query); then, for each of the four commitment trees, one SinglePathMerkleProofVar::verify per query in
 * TRANSCRIPT query order (components/recursive/answer/src/lib.rs:214-258, data_structures/src/lib.rs:315-354); then one
 * SinglePairMerkleProofVar::verify per query for the first FRI layer and for every inner layer
 * (folding/src/lib.rs:23-33,186-189, data_structures/src/lib.rs:400-464).  What a next-level prover pads to a multiple of
 * 16 and proves as its Poseidon component (six trace rows per record).  Any mix of shapes; the records of a proof
 * that is rejected behind the parser are whatever its (failing) verification computed.  With d_flow the per-query
 * kernels walk every path to the root themselves (no shared top-of-tree cap, no shared row hashes), like the circuit.
 * PINNED value by value through the fixture chain: the circuit that verifies fixture K, rebuilt from these records and the
 * witness below, evaluated at fixture K+1's OODS point, gives K+1's sampled values (tests/test_witness_gpu.py: the GPU's
 * own flow reproduces the next fixture's 88 Poseidon columns for all 14 pairs; oracle/recursion_circuit).  The wire
 * indices of PoseidonEntry / SwapOption::addr are constants of the shape: rsv_witness_program_export (flow_wires). */
#define RSV_TRANSCRIPT_WORDS 284
typedef struct {
    uint32_t n_queries, max_log, n_inner; /* declared common shape; ignored when no path output is requested */
    uint32_t* d_transcript;
    uint32_t* d_trace_sib;
    uint32_t* d_trace_pos;
    uint32_t* d_trace_cols;
    uint32_t* d_fri_sib;
    uint32_t* d_fri_cols;
    uint32_t* d_fri_folded;
    uint32_t* d_query_values;
    /* PoseidonFlow (ABI v3), see above */
    uint32_t* d_flow;
    uint8_t* d_flow_swap;
    uint32_t* d_flow_count;
    uint32_t flow_stride;
    /* the batch's accept bitmap and count from the verifying pass (ABI v3): what rsv_accept_bitmap_dev computes from
     * the accept bytes, written by the kernel that writes the verdicts (one launch less on the latency path of a
     * small batch).  d_accept_bitmap: ceil(n/32) words; d_accept_count: optional u64, needs d_accept_bitmap. */
    uint32_t* d_accept_bitmap;
    uint64_t* d_accept_count;
} rsv_hints_out;
int rsv_verify_hints_dev(rsv_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                         const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, const rsv_hints_out* out, uint8_t* d_accept,
                         uint8_t* d_reason);
/* Same with every pointer (blob, offsets, the outputs named in *out, accept, reason) in HOST memory: the library
 * stages them through HBM.  rsv_trace_paths, rsv_fri_paths and rsv_transcript_batch are special cases of it. */
int rsv_verify_hints(const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg,
                     const rsv_public_input* pi, size_t n_pi,
                     const rsv_hints_out* out, uint8_t* accept, uint8_t* reason, int device);
/* Records in the PoseidonFlow of one proof with these component log sizes under cfg (pure arithmetic, no device). */
int rsv_poseidon_flow_count(uint32_t log_size_plonk, uint32_t log_size_poseidon, const rsv_pcs_config* cfg, uint32_t* count);
/* Host-buffer convenience for the transcript rows only (any mix of shapes): out is [n][RSV_TRANSCRIPT_WORDS]. */
int rsv_transcript_batch(const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg, uint32_t* out,
                         int device);

/* ---- SURVEY 8f.1, completed: the recursion circuit's witness ------------------------------------------------------
 * What the reference's circuit leaves for the next prover is `variables: Vec<QM31>`
 * (constraint_system/src/plonk_with_poseidon.rs:19): one value per cs.add / cs.mul / cs.mul_constant / new_m31 /
 * new_qm31 call of the gadgets, in call order (:140-283); the trace columns a_val / b_val / c_val are that vector read
 * through the wires (:571-618).  The reference computes it by running the gadgets on one proof at a time on the CPU.
 * Which gate or hint produces variable k is the same for every proof of one shape (statement sizes + PcsConfig), so
 * here the gadgets are run once per shape on the host (rsv_witness_program_build: csrc/circuit_*.hpp, a mirror of the
 * reference's ConstraintSystemRef / M31Var / ... / FiatShamirResults / CompositionCheck / AnswerResults / FoldingResults
 * that writes down what it did) and the resulting PROGRAM — one instruction per variable, sorted by dependency depth — is evaluated
 * on the GPU for a whole batch, from the hints of the verifying pass (proof words, PoseidonFlow records, per-query
 * column values).  The circuit is `copies` copies of the verifier in one constraint system, as
 * examples/multi-proofs/src/main.rs:66-139 builds it (`multipliers`); every copy verifies the same proof.
 *
 * Instruction = 8 u32: op, dst, a, b, imm0..imm3 (ops: recursive-stwo_amd/csrc/k_witness.hpp WitnessOp; the table with
 * their meaning heads recursive-stwo_amd/witness_program.py).  rsv_witness_program_create checks every index the device
 * will use (RSV_E_RANGE otherwise) and keeps a copy in HBM. */
typedef struct rsv_witness_program rsv_witness_program;
typedef struct {
    uint32_t log_size_plonk, log_size_poseidon;            /* the statement of the proofs the program applies to */
    uint32_t pow_bits, log_blowup, log_last, n_queries;    /* their PcsConfig */
    uint32_t n_inner;                                      /* FRI inner layers */
    uint32_t flow_count;                                   /* rsv_poseidon_flow_count of the shape */
    uint32_t copies;                                       /* copies of the verifier in the circuit */
} rsv_witness_shape;
int rsv_witness_program_create(const uint32_t* instr, size_t n_instr, const uint32_t* level_offsets, size_t n_levels,
                               uint32_t n_vars, const rsv_witness_shape* shape, int device, rsv_witness_program** out);
void rsv_witness_program_destroy(rsv_witness_program* prog);
/* The program of the shape of `proof` (a template: any proof of that shape that verifies under cfg with these public
 * inputs, else RSV_E_RANGE), for a circuit of `copies` copies of the verifier: runs the library's mirror of the
 * reference's gadgets (csrc/circuit_{cs,gadgets,verifier}.hpp: ConstraintSystemRef, M31Var .. QM31Var, BitsVar,
 * Poseidon2HalfVar, ChannelVar, the Merkle hasher, circle points, LinePolyVar, query positions; PlonkWithPoseidonProofVar,
 * FiatShamirResults, CompositionCheck, AnswerResults, FoldingResults) once over the template on the host, fed with the
 * hints of the GPU's verifying pass over it.  A few milliseconds per 50 000 variables, plus that one verifying pass.
 * set_walks (NULL = all zero): one byte per copy.  AnswerResults::compute walks two std HashSet<isize> = {0, -1} whose
 * order Rust seeds per process (components/recursive/answer/src/lib.rs:44-71), so the reference's own circuit comes in
 * four variants per copy that differ in the ORDER of two pairs of blocks of variables (the values are the same); bit 0 =
 * the Plonk set is walked -1 first, bit 1 = the Poseidon set.  A prover that already fixed its gate list (preprocessed
 * columns) passes the walks that list was made with.  A PROVER INTEGRATION MUST DO SO: the library cannot observe which of
 * the four orders a given run of the reference took (the tests recover them per fixture pair by trying the four against the
 * next fixture's sampled values, tests/pin_recursion_circuit.py); with the wrong walks every value is still right but two
 * pairs of blocks of `variables` sit in the other order than the prover's wires expect. */
int rsv_witness_program_build(const uint8_t* proof, size_t len, const rsv_pcs_config* cfg, const rsv_public_input* pi, size_t n_pi,
                              uint32_t copies, const uint8_t* set_walks, int device, rsv_witness_program** out);
/* Sizes, then the arrays (any pointer may be NULL): instr [n_vars][8], level_offsets [n_levels + 1], flow_wires
 * [copies * flow_count][5] = PoseidonEntry::wire of r1..r4 and SwapOption::addr of every invocation (constants of the
 * shape; known to built programs only, RSV_E_SIZE otherwise) — what rsv_witness_program_create takes back. */
int rsv_witness_program_info(const rsv_witness_program* prog, uint32_t* n_vars, uint32_t* n_levels, rsv_witness_shape* shape);
int rsv_witness_program_export(const rsv_witness_program* prog, uint32_t* instr, uint32_t* level_offsets, uint32_t* flow_wires);
/* The circuit's gate list as the gadgets left it (built programs only, RSV_E_SIZE otherwise): gates [n_rows][6] =
 * a_wire, b_wire, c_wire, op, poseidon_wire, enforce_c_m31 per Plonk row (plonk_with_poseidon.rs:23-36, before pad()) —
 * with `variables` and the flow everything generate_plonk_with_poseidon_circuit (:522-629) and populate_logup_arguments
 * (:345-466) read.  The rows are constants of the shape EXCEPT `op` at the rows listed in witness_ops [n][3] = (row, bit
 * variable, constant): CirclePointM31Var::select takes its gate constant from the selected value
 * (primitives/circle/src/lib.rs:83-98), so there op = constant where the proof's variables[bit] is 1 and 0 where it is 0;
 * `gates` holds the template's.  Call with NULL arrays for the two counts first. */
int rsv_witness_program_gates(const rsv_witness_program* prog, uint32_t* n_rows, uint32_t* n_witness_ops, uint32_t* gates,
                              uint32_t* witness_ops);
/* HBM the context will hold for a batch of n proofs (hints of the verifying pass + variables[var][proof]). */
int rsv_witness_scratch_bytes(const rsv_witness_program* prog, size_t n, size_t* bytes);
/* Verifies the batch (as rsv_verify_hints_dev, under cfg = the program's single configuration, else RSV_E_SIZE) and
 * writes d_variables [n][n_vars][4]: row i = the `variables` vector of the circuit that verifies proof i
 * ([n_vars][n][4] under RSV_OPT_WITNESS_LAYOUT = 2).
 * d_accept[i] = 1 iff proof i verified AND is of the program's shape; only those rows are defined (the others are
 * unspecified).  d_variables 16-byte aligned.
 * d_flow [n][flow_count][32] + d_flow_swap [n][flow_count] (optional, both or neither; as rsv_hints_out::d_flow with
 * flow_stride = the shape's flow_count): the PoseidonFlow of ONE copy of the verifier — the other thing the next prover
 * needs; every copy invokes the same permutations, the wire indices of invocation k of copy c are the host's
 * (rsv_witness_program_export, flow_wires).  Without them the records live in the context's scratch only. */
int rsv_witness_eval_dev(rsv_ctx* ctx, const rsv_witness_program* prog, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                         const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, uint32_t* d_variables, uint32_t* d_flow,
                         uint8_t* d_flow_swap, uint8_t* d_accept, uint8_t* d_reason);
/* Same on host buffers. */
int rsv_witness_eval(const rsv_witness_program* prog, const uint8_t* blob, const uint64_t* offsets, size_t n, const rsv_cfg_set* cfg,
                     const rsv_public_input* pi, size_t n_pi, uint32_t* variables, uint32_t* flow, uint8_t* flow_swap, uint8_t* accept,
                     uint8_t* reason, int device);

/* Pack n accept bytes (device) into a little-endian bitmap of ceil(n/32) u32
 * words (device) and return the popcount through *d_count (device u64, may be NULL).
 * This is the buffer the multi-GPU host exchanges with one RCCL all-gather (rsv_exchange_run, below). */
int rsv_accept_bitmap_dev(rsv_ctx* ctx, const uint8_t* d_accept, size_t n,
                          uint32_t* d_bitmap, uint64_t* d_count);

/* ---- e: more than one GPU (BASELINE configs[3]) -----------------------------------------------------------
 * Proofs are independent, so a job shards by contiguous index range and NOTHING is exchanged while it verifies; what is
 * left is to put the shards' accept bits together.  Two layouts, both behind this header:
 *
 * (1) ONE process drives N devices — what the reference's driver is (examples/multi-proofs/src/main.rs:198-295: one
 *     process walks the whole chain).  rsv_multi holds one context per entry of `devices` (a device may be named more
 *     than once: several contexts on one GPU) and runs one host thread per context per call.  A process that owns every
 *     shard has nobody to send to, so the job's accept bytes / bitmap / count are assembled on the host: no collective.
 * (2) One process per GPU (torchrun, MPI, N copies of a Rust binary).  Each rank verifies its shard and the ranks
 *     exchange their bitmap slices with ONE ncclAllGather and their counts with ONE ncclAllReduce over RCCL / xGMI
 *     (rsv_exchange).  RCCL is bound at run time, never linked; RSV_E_UNAVAILABLE where it cannot be loaded.
 *
 * Shard rules (both layouts, and recursive-stwo_amd/sharding.py).  Shards are contiguous index ranges, shard 0 first.
 *   rsv_shard_range   a UNIFORM job (proofs of one shape, or shapes in round-robin order): rank r of `world` owns
 *                     [lo, hi) with sizes differing by at most one, the larger shards first.
 *   rsv_shard_plan    any job, balanced by WORK: bytes are the work (21 - 26.5 proof bytes per permutation over every shape
 *                     of the reference, and the verifier is permutation-bound), so the cut between ranks r - 1 and r is
 *                     the proof boundary nearest to r / world of the job's bytes.  The reference's own job arrives ordered
 *                     by level (examples/multi-proofs/src/main.rs:198-295: 435 KB / 80-query proofs first, 76 KB / 8-query
 *                     ones last): cut by count, rank 0 of 8 gets 4.5 x the bytes of rank 7; cut by bytes, every rank the
 *                     same within one proof.  lens: n proof lengths (HOST); lo, hi: `world` entries each (HOST, written);
 *                     world in 1 .. 4096 (RSV_E_SIZE).  Deterministic: every rank computes the same plan from the same
 *                     lengths.  rsv_multi_verify_batch_host cuts its job this way. */
void rsv_shard_range(size_t n_total, size_t rank, size_t world, size_t* lo, size_t* hi);
int rsv_shard_plan(const uint64_t* lens, size_t n, size_t world, size_t* lo, size_t* hi);

typedef struct rsv_multi rsv_multi;
/* n_devices in 1 .. 64; every entry a valid HIP device index (RSV_E_DEVICE otherwise). */
int rsv_multi_create(const int* devices, size_t n_devices, rsv_multi** out);
void rsv_multi_destroy(rsv_multi* m);
size_t rsv_multi_size(const rsv_multi* m);
/* Context of rank r (owned by m): for rsv_ctx_set_option, or to use a rank on its own.  NULL if out of range. */
rsv_ctx* rsv_multi_ctx(rsv_multi* m, size_t rank);
/* The whole job in HOST memory, as the reference's caller holds it (one serialized buffer per proof): rank r runs
 * rsv_verify_batch_host on shard rsv_shard_range(n, r, size) from its own thread (gather -> pinned -> DMA -> verify,
 * pipelined per device).  accept / reason: n bytes each (reason may be NULL); bitmap: ceil(n / 32) little-endian words
 * (bit i = accept[i]) or NULL; count: accepted proofs or NULL.  cfg->cfg_of (HOST memory here) indexes the whole job.
 * Blocks until every verdict is written.  The first failing rank's status is returned (RSV_E_NOMEM: a rank could not
 * allocate its host staging). */
int rsv_multi_verify_batch_host(rsv_multi* m, const uint8_t* const* proofs, const uint64_t* lens, size_t n,
                                const rsv_cfg_set* cfg, const rsv_public_input* pi, size_t n_pi, uint8_t* accept,
                                uint8_t* reason, uint32_t* bitmap, uint64_t* count);
/* The job already resident in HBM, shard r on the device of context r (the caller chose the split; the job's proof
 * order is shard 0, shard 1, ...).  Pointers are DEVICE pointers of that device, as for rsv_verify_batch_dev;
 * d_cfg_of / d_accept / d_reason may be NULL (no per-proof configuration index / verdict bytes not wanted). */
typedef struct rsv_shard {
    const uint8_t* d_blob;
    const uint64_t* d_offsets;
    size_t n;
    const uint8_t* d_cfg_of;
    uint8_t* d_accept;
    uint8_t* d_reason;
} rsv_shard;
/* n_shards must equal rsv_multi_size(m).  Every context verifies its shard (one pass that also packs the shard's
 * bitmap and count, rsv_hints_out::d_accept_bitmap), the slices come back over PCIe (n / 8 bytes) and are placed at
 * their bit offsets: bitmap = ceil(sum n / 32) HOST words or NULL, count = HOST u64 or NULL.  cfg->cfg_of is ignored
 * (each shard has its own d_cfg_of).  Blocks until all contexts are done. */
int rsv_multi_verify_batch_dev(rsv_multi* m, const rsv_shard* shards, size_t n_shards, const rsv_cfg_set* cfg,
                               const rsv_public_input* pi, size_t n_pi, uint32_t* bitmap, uint64_t* count);

/* Layout (2).  TESTED ON ONE DEVICE ONLY so far (world 1 on hardware, world 2 / 3 / 8 over gloo on CPUs: this pool has
 * one-GPU boxes).  rank 0 obtains an id (ncclGetUniqueId) and hands its RSV_EXCHANGE_ID_BYTES bytes to the other ranks by
 * whatever channel the host has (a file, an environment variable, MPI, a torch store); every rank then calls
 * rsv_exchange_create — collectively: it returns when all `world` ranks have joined (ncclCommInitRank on ctx's device).
 * Per batch: verify the shard with rsv_verify_hints_dev asking for d_accept_bitmap = d_local (a buffer of slice_words
 * words; the verifying pass writes ceil(shard / 32) of them with the bits above the shard zero) and d_accept_count =
 * d_count, then rsv_exchange_run — enqueued on ctx's stream behind the verifying pass, no host synchronisation: it zeroes
 * the words of d_local above the shard's own (a narrower shard than the widest sends slice_words words all the same),
 * afterwards d_gathered [world][slice_words] holds every rank's slice — nothing but accept bits and zeros — and *d_count
 * the job's total on every rank.  rsv_exchange_assemble (pure host arithmetic, no RCCL needed) turns a host copy of
 * d_gathered into the job's accept bytes and / or contiguous bitmap.
 * rsv_exchange_create cuts the job with rsv_shard_range; rsv_exchange_create_plan / rsv_exchange_assemble_plan take the
 * cuts of every rank (lo, hi: `world` entries each, contiguous from 0 — rsv_shard_plan's, or the caller's own;
 * RSV_E_SIZE otherwise), slice_words is then the widest shard's.
 * Lifetime: destroy an exchange BEFORE the context it was created on (it enqueues on that context's stream). */
#define RSV_EXCHANGE_ID_BYTES 128
typedef struct rsv_exchange rsv_exchange;
int rsv_exchange_available(void);     /* 1 if RCCL could be bound in this process, else 0 */
int rsv_exchange_rccl_version(void);  /* ncclGetVersion, 0 if unavailable */
int rsv_exchange_unique_id(uint8_t* id128);
int rsv_exchange_create(rsv_ctx* ctx, const uint8_t* id128, int rank, int world, size_t n_total, rsv_exchange** out);
int rsv_exchange_create_plan(rsv_ctx* ctx, const uint8_t* id128, int rank, int world, const size_t* lo, const size_t* hi,
                             rsv_exchange** out);
void rsv_exchange_destroy(rsv_exchange* x);
/* This rank's [lo, hi) and the slice size every rank contributes (the largest shard's bitmap words, at least 1). */
int rsv_exchange_layout(const rsv_exchange* x, size_t* lo, size_t* hi, size_t* slice_words);
int rsv_exchange_run(rsv_exchange* x, uint32_t* d_local, uint32_t* d_gathered, uint64_t* d_count);
int rsv_exchange_assemble(size_t n_total, size_t world, const uint32_t* gathered, uint8_t* accept, uint32_t* bitmap);
int rsv_exchange_assemble_plan(size_t world, const size_t* lo, const size_t* hi, const uint32_t* gathered, uint8_t* accept,
                               uint32_t* bitmap);

/* Per-stage kernel time of the last rsv_verify_batch_dev on this ctx, measured
 * with HIP events on the ctx stream (ms); all zero unless RSV_OPT_STAGE_TIMES = 1 was set
 * on the context before the call.  names[i] are static strings.
 * Returns the number of stages written (<= cap).  Synchronises the stream. */
int rsv_last_stage_times(rsv_ctx* ctx, const char** names, float* ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* RSV_H_ */
