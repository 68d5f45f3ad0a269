// layout.hpp — HBM data layout shared by the verify kernels and the host launcher.
//
// A batch is the caller's blob (proof i = blob[offsets[i], offsets[i+1])) plus a
// per-proof workspace record written by the early stages and read by the
// later ones.  All "offsets" inside the records are in 32-bit WORDS relative to
// the start of the proof (every field of the bincode encoding is 4-byte
// aligned).
#pragma once
#include <stdint.h>

// (this header and the host_*.hpp / circuit_*.hpp ones are plain C++: g++ builds them for the host-logic sanitizer pass,
// tests/host_logic_asan.cpp, where the HIP qualifiers mean nothing)
#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

namespace rsv {

constexpr int MAXQ = 128;      // max FRI queries per proof (fixtures: 8..80)
constexpr int MAX_INNER = 28;  // max FRI inner layers (M <= 30)
constexpr int MAX_LOG = 30;    // max column log size M
constexpr int N_SAMPLES = 142; // 50+60+16+8 columns, 8 of them with two mask points
constexpr int N_APOW = 136;    // alpha powers kept for the quotient numerators

// Column counts per commitment tree: plonk | poseidon component.
__host__ __device__ constexpr uint32_t plonk_cols(int t) { return t == 0 ? 10u : t == 1 ? 12u : t == 2 ? 8u : 0u; }
__host__ __device__ constexpr uint32_t poseidon_cols(int t) { return t == 0 ? 40u : t == 1 ? 48u : t == 2 ? 8u : 0u; }
__host__ __device__ constexpr uint32_t tree_cols(int t) { return t == 3 ? 8u : plonk_cols(t) + poseidon_cols(t); }
// Mask sizes: one sample per column except interaction columns 4-7 and 12-15 ([-1, 0]).
__host__ __device__ constexpr uint32_t n_samples_of(int t, int c) { return (t == 2 && (c & 4)) ? 2u : 1u; }

// Fixed word offsets of the constant-shape prefix of a proof.
constexpr uint32_t W_LP = 0, W_LQ = 1, W_PLONK_SUM = 2, W_POSEIDON_SUM = 6, W_POW_BITS = 10, W_BLOWUP = 11,
                   W_LOG_LAST = 12, W_NQ = 13, W_NCOMMIT = 15, W_COMMIT0 = 17, W_NTREES = 49;

struct SampleTable {
    uint16_t off[N_SAMPLES];  // word offset of each sampled value (tree-major, column-major, sample-minor)
    uint16_t col_prefix[134]; // word offset of each column's u64 sample-count prefix
    uint16_t tree_prefix[4];  // word offset of each tree's u64 column-count prefix
    uint16_t end;             // first word after sampled_values (= decommitments prefix)
    uint16_t pre_of[N_SAMPLES];  // sample -> word offset of its column's u64 sample-count prefix if it is the column's FIRST sample, else 0
    uint8_t cnt_of[N_SAMPLES];   // ... and the count that prefix must hold (1 or 2)
};
__host__ __device__ constexpr SampleTable make_sample_table() {
    SampleTable t{};
    uint32_t pos = W_NTREES + 2, k = 0, c_all = 0;
    for (int tr = 0; tr < 4; tr++) {
        t.tree_prefix[tr] = (uint16_t)pos;
        pos += 2;
        for (uint32_t c = 0; c < tree_cols(tr); c++) {
            t.col_prefix[c_all++] = (uint16_t)pos;
            t.pre_of[k] = (uint16_t)pos;
            t.cnt_of[k] = (uint8_t)n_samples_of(tr, (int)c);
            pos += 2;
            for (uint32_t s = 0; s < n_samples_of(tr, (int)c); s++) {
                t.off[k++] = (uint16_t)pos;
                pos += 4;
            }
        }
    }
    t.end = (uint16_t)pos;
    return t;
}
// index of the first sample of each tree inside the flattened sample list
constexpr int S_T0 = 0, S_T1 = 50, S_T2 = 110, S_T3 = 134;

struct FriLayerRef {
    uint32_t wit_off, wit_n;    // fri_witness: wit_n QM31 values
    uint32_t hash_off, hash_n;  // decommitment.hash_witness: hash_n hashes of 8 words
    uint32_t commit_off;        // layer commitment (8 words)
};

// Written by k_parse.
struct ProofMeta {
    uint32_t reason;  // RSV_R_OK or RSV_R_PARSE
    uint32_t n_words;
    uint32_t lp, lq, pow_bits, blowup, log_last, nq, n_inner, A, B, M;
    uint32_t hw_off[4], hw_n[4];  // trace-tree hash witnesses
    uint32_t qv_off[4], qv_n[4];  // trace-tree queried values
    uint32_t nonce_off;
    FriLayerRef first;
    FriLayerRef inner[MAX_INNER];
    uint32_t last_off, last_n;
};

// Failure flags accumulated by the stages: bit r is set when stage rsv_reason r failed.
// k_finalize reports the lowest set bit (= the order in which the reference's
// stages would have panicked).

struct QBatch {
    uint32_t sa[4], sb[4];            // sum of line coefficients a_k, b_k over the batch (QM31)
    uint32_t prx[2], pix[2], pry[2], piy[2];  // sample point: x = prx + pix*u, y = pry + piy*u (CM31 each)
};

// Canonicity (every field-element word of a proof must be < P, else RSV_R_PARSE) is checked WHERE THE WORDS ARE READ:
// the transcript checks what it absorbs (total sums, commitments, sampled values, FRI commitments, last-layer
// coefficients), k_row_hash the queried values, k_query the FRI witness values, the Merkle kernels the hash witnesses;
// every other word is a length prefix or header word that k_parse pins to a small value.  A non-canonical word
// raises bit RSV_R_PARSE of ProofCtx::flags (the lowest reason, so it outranks every later stage, as the dedicated
// scan of rounds 1-2 did).  A witness list is read completely exactly when the stage that consumes it finds its
// length right; a stage that finds it wrong (the proof is rejected there anyway) also raises F_RESCAN, and k_rescan
// then reads that proof once in full, so that the REASON is the defined one (PARSE if any word anywhere is non-canonical).
constexpr uint32_t F_RESCAN = 1u << 30;
constexpr uint32_t F_REASON_MASK = (1u << 13) - 1u;

// Written by k_transcript / k_plan, read by the per-query kernels.
struct ProofCtx {
    uint32_t flags;
    uint32_t flow_on;        // PoseidonFlow emission: 1 when this proof's records fit the caller's stride (k_transcript)
    uint32_t front_over;     // split transcript: the front half saw a non-canonical word (it may not touch `flags`,
                             // which the parser zeroes while the front half runs)
    uint32_t n_sizes;
    uint32_t sizes[3];       // distinct column log sizes, descending (M first)
    uint32_t fw_base[3];     // first-layer fri_witness base index per size
    uint32_t z[4], alpha[4], rc[4], oods_t[4], oods_x[4], oods_y[4], after[4];
    uint32_t pow_digest[8];
    uint32_t fri_alpha[MAX_INNER + 1][4];
    uint32_t raw_q[MAXQ];    // raw query words in transcript order
    uint32_t q[MAXQ];        // query positions at log M, ascending
    uint8_t qperm[MAXQ];     // qperm[k] = transcript index of the k-th smallest query
    QBatch batch[3][2];      // per size group: batch 0 = OODS point, batch 1 = OODS - trace step
    uint32_t n_batches[3];
    uint32_t apow[N_APOW][4];  // -2u * after^k
};

// ---- PoseidonFlow (include/rsv.h: rsv_hints_out::d_flow) -------------------------------------------
// One record per Poseidon2HalfVar::permute invocation of the circuit that verifies the proof, in the circuit's
// invocation order (constraint_system/src/plonk_with_poseidon.rs:117-128; order: examples/multi-proofs/src/main.rs:69-139):
//   [transcript][tree 0: one path per query, TRANSCRIPT query order][tree 1][tree 2][tree 3]
//   [FRI first layer: one pair path per query][inner layer 0: one per query] ... [inner layer n_inner - 1]
// The position of every record follows from the proof's shape alone, so each kernel writes the records of the
// permutations it executes at their final index; nothing is appended.
__host__ __device__ constexpr uint32_t flow_chunks(uint32_t n_cols) { return (n_cols + 7u) / 8u; }
// FiatShamirResults::compute (components/recursive/fiat_shamir/src/lib.rs:44-130): 4 mixes + draw, mix + mix + draw,
// mix + draw, 71 mixes + draw, (mix + draw) per FRI layer, the last-layer polynomial two felts per mix, the nonce,
// ceil(n_queries / 4) draws (the circuit draws two felts per four queries and truncates).
__host__ __device__ constexpr uint32_t flow_transcript_len(uint32_t nq, uint32_t n_inner, uint32_t last_n) {
    return 82u + 2u * (1u + n_inner) + (last_n + 1u) / 2u + 1u + (nq + 3u) / 4u;
}
// SinglePathMerkleProofVar::verify (components/recursive/data_structures/src/lib.rs:315-354): leaf sponge + rate,
// one swap-permute per level, and at the lower column level the column sponge and the combine.
__host__ __device__ constexpr uint32_t flow_trace_path_len(int t, uint32_t A, uint32_t B, uint32_t M) {
    const uint32_t mx = t == 3 ? M : (A > B ? A : B);
    const uint32_t nc_leaf = t == 3 ? 8u : ((A == mx ? plonk_cols(t) : 0u) + (B == mx ? poseidon_cols(t) : 0u));
    const uint32_t nc_lower = (t == 3 || A == B) ? 0u : (A < B ? plonk_cols(t) : poseidon_cols(t));
    return flow_chunks(nc_leaf) + 1u + mx + (nc_lower ? flow_chunks(nc_lower) + 1u : 0u);
}
// SinglePairMerkleProofVar::verify (data_structures/src/lib.rs:400-464): 2 + 2 leaf permutations, one swap-permute per
// level, and four more at every lower column level of the first-layer tree (two column capacities, two combines).
__host__ __device__ constexpr uint32_t flow_pair_path_len(uint32_t s, uint32_t A, uint32_t B, uint32_t M) {
    return s == 0 ? 4u + M + 4u * (A == B ? 1u : 2u) : 4u + (M - s);
}
__host__ __device__ constexpr uint32_t flow_trace_base(int t, uint32_t nq, uint32_t n_inner, uint32_t last_n, uint32_t A, uint32_t B,
                                                       uint32_t M) {
    uint32_t at = flow_transcript_len(nq, n_inner, last_n);
    for (int k = 0; k < t; k++) at += nq * flow_trace_path_len(k, A, B, M);
    return at;
}
__host__ __device__ constexpr uint32_t flow_pair_base(uint32_t s, uint32_t nq, uint32_t n_inner, uint32_t last_n, uint32_t A, uint32_t B,
                                                      uint32_t M) {
    uint32_t at = flow_trace_base(4, nq, n_inner, last_n, A, B, M);
    for (uint32_t k = 0; k < s; k++) at += nq * flow_pair_path_len(k, A, B, M);
    return at;
}
__host__ __device__ constexpr uint32_t flow_total(uint32_t nq, uint32_t n_inner, uint32_t last_n, uint32_t A, uint32_t B, uint32_t M) {
    return flow_pair_base(1u + n_inner, nq, n_inner, last_n, A, B, M);
}
constexpr uint32_t FLOW_WORDS = 32;  // left8 | right8 | out_rate8 | out_cap8 (one 128-byte line); the swap bit is a byte of its own

// Per-proof decommitment plan (k_plan), indexed by tree level l = 0..M and lane j
// (lane j owns the j-th smallest query):
//   ent[l*G + j]  = RB | LB<<8 | SIB<<16
//       RB  = number of distinct nodes at level l to the left of lane j's node
//       LB  = number of distinct parents (level l-1) to the left of lane j's
//             whose other child is not on any query path (=> hash witness)
//       SIB = lane that owns the sibling node at level l, 0xFF if it comes from the witness
//   lvl[l]        = ND | TL<<8 | S<<16
//       ND = distinct nodes at level l, TL = parents at level l-1 lacking a child,
//       S  = sum of TL over levels >= l
struct PlanHdr {
    uint32_t lvl[MAX_LOG + 2];
    // FRI first-layer tree (column data at levels sizes[0..n_sizes)):
    uint16_t wf[MAX_LOG + 2];  // hash-witness base index of the step from child level l to l-1
    uint16_t wf_total, pad;
};
// fl[(d*G + j)] for the (up to two) non-leaf data levels d of the first-layer tree:
//   w_self | w_sib << 16   (0xFFFF = none / sibling owned by lane SIB)

// One instruction of a witness program (rsv_witness_program_*, k_witness.hpp) = 8 words: op, dst, a, b, imm0..3.
enum WitnessOp : uint32_t {
    W_CONST, W_ADD, W_MUL, W_MULC, W_COPY, W_INV, W_INV0, W_QINV, W_CINV, W_COORD, W_BIT, W_FLOW, W_WORD, W_WORD4,
    W_FRI_COMMIT, W_LAST_POLY, W_NONCE, W_TRACE_COL, W_FRI_COL, W_N_OPS
};

}  // namespace rsv
