// k_witness.hpp — the recursion circuit's `variables` vector for a batch of proofs of one shape (rsv_witness_eval_dev).
// The reference fills the vector while it runs the circuit's gadgets on ONE proof
// (constraint_system/src/plonk_with_poseidon.rs:140-283).  Which gate or hint produces variable k is the same for every
// proof of a shape, so the host writes that down once as a program (rsv_witness_program_build, circuit_builder.inc: one
// instruction per variable, sorted by dependency depth) and this kernel evaluates one LEVEL of it per launch over
// (instructions of the level) x (proofs).  Variables live in HBM as vars[variable][proof] (QM31 = 16 B), so a wave reads
// and writes 1 KB rows; the work is HBM streaming (3 x 16 B per instruction and proof), not arithmetic.
// Hint instructions read what rsv_verify_hints_dev left behind: the proof words (through the parser's offsets), the
// PoseidonFlow records (the permutation outputs are hints of the circuit), the per-query column values of the four
// commitment trees and of the FRI trees.
#pragma once
#include "verify_common.hpp"

namespace rsv {

// (enum WitnessOp: layout.hpp)
struct WitnessArgs {
    const uint32_t* instr;  // [n_instr][8]: op, dst, a, b, imm0..3
    uint32_t begin, end;    // this level's instructions
    uint32_t n;             // proofs
    uint4* vars;            // [n_vars][n]
    const uint8_t* blob;
    const uint64_t* offsets;
    const ProofMeta* metas;
    const uint8_t* accept;
    const uint4* flow;        // [n][flow_stride][8]  (32 words per record)
    uint32_t flow_stride;
    const uint32_t* trace_cols;  // [n][4][nq][64]
    const uint32_t* fri_cols;    // [n][n_trees][nq][3][8]
    uint32_t nq, n_trees;
};

__device__ __forceinline__ QM31 q_of(uint4 v) { return q_mk(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 u4_of(QM31 q) { return make_uint4(q.a.a, q.a.b, q.b.a, q.b.b); }
__device__ __forceinline__ uint4 u4_words(const uint32_t* w) { return make_uint4(w[0], w[1], w[2], w[3]); }

// one instruction (its eight words: w0 = op, dst, a, b; w1 = imm0..3) for one proof
__device__ __forceinline__ void witness_exec(const WitnessArgs& a, const uint4 w0, const uint4 w1, uint32_t p) {
    const uint32_t op = w0.x, dst = w0.y, x = w0.z, y = w0.w, i0 = w1.x, i1 = w1.y, i2 = w1.z;
    const size_t n = a.n;
    uint4 r = make_uint4(0, 0, 0, 0);
    const bool ok = a.accept[p] != 0;  // a rejected proof's sections may not be where the program expects them
    switch (op) {
    case W_CONST: r = w1; break;
    case W_ADD: r = u4_of(q_add(q_of(a.vars[x * n + p]), q_of(a.vars[y * n + p]))); break;
    case W_MUL: r = u4_of(q_mul(q_of(a.vars[x * n + p]), q_of(a.vars[y * n + p]))); break;
    case W_MULC: r = u4_of(q_mul_m(q_of(a.vars[x * n + p]), i0)); break;
    case W_COPY: r = a.vars[x * n + p]; break;
    case W_INV: r.x = m_inv(a.vars[x * n + p].x); break;  // 0 -> 0
    case W_INV0: r.x = m_inv(a.vars[x * n + p].x); break;
    case W_QINV: {
        const uint4 v = a.vars[x * n + p];
        if (v.x | v.y | v.z | v.w) r = u4_of(q_inv(q_of(v)));
        break;
    }
    case W_CINV: {
        const uint4 v = a.vars[x * n + p];
        if (v.x | v.y) {
            const CM31 c = c_inv(c_mk(v.x, v.y));
            r.x = i0 ? c.b : c.a;
        }
        break;
    }
    case W_COORD: {
        const uint4 v = a.vars[x * n + p];
        r.x = i0 == 0 ? v.x : i0 == 1 ? v.y : i0 == 2 ? v.z : v.w;
        break;
    }
    case W_BIT: r.x = (a.vars[x * n + p].x >> i0) & 1u; break;
    case W_FLOW:
        if (ok) r = a.flow[((size_t)p * a.flow_stride + i0) * 8 + (i1 >> 2)];
        break;
    default: {
        if (!ok) break;
        const ProofMeta& m = a.metas[p];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(a.blob + a.offsets[p]);
        switch (op) {
        case W_WORD: r.x = w[i0]; break;
        case W_WORD4: r = u4_words(w + i0); break;
        case W_FRI_COMMIT: r = u4_words(w + (i0 == 0 ? m.first.commit_off : m.inner[i0 - 1].commit_off) + 4 * i1); break;
        case W_LAST_POLY: r = u4_words(w + m.last_off + 4 * i0); break;
        case W_NONCE: {  // the 22 / 21 / 21-bit split of data_structures/src/lib.rs:197-213
            const uint64_t nonce = (uint64_t)w[m.nonce_off] | ((uint64_t)w[m.nonce_off + 1] << 32);
            r.x = i0 == 0 ? (uint32_t)(nonce & ((1u << 22) - 1)) : (uint32_t)((nonce >> (i0 == 1 ? 22 : 43)) & ((1u << 21) - 1));
            break;
        }
        case W_TRACE_COL: r.x = a.trace_cols[(((size_t)p * 4 + i0) * a.nq + i1) * 64 + i2]; break;
        case W_FRI_COL: r = u4_words(a.fri_cols + (((size_t)p * a.n_trees + i0) * a.nq + i1) * 24 + i2); break;
        default: break;
        }
    }
    }
    a.vars[dst * n + p] = r;
}
__device__ __forceinline__ void witness_exec(const WitnessArgs& a, uint32_t i, uint32_t p) {
    const uint4* in = reinterpret_cast<const uint4*>(a.instr) + (size_t)i * 2;
    witness_exec(a, in[0], in[1], p);
}

// one level, grid = (instructions of the level) x (proofs)
__global__ __launch_bounds__(256) void k_witness_level(WitnessArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = a.begin + (uint32_t)(t / a.n), p = (uint32_t)(t % a.n);
    if (i >= a.end) return;
    witness_exec(a, i, p);
}

// the same for batches of at least a wave of proofs: blockIdx.y = the instruction, so that the instruction words are scalar
// loads and the switch a scalar branch (in the 1-D form a wave may straddle two instructions, and every lane divides)
__global__ __launch_bounds__(256) void k_witness_level_wide(WitnessArgs a) {
    const uint32_t i = a.begin + blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    witness_exec(a, i, p);
}

// Levels l0 .. l1 walked by ONE workgroup for its own proofs (dependencies never cross proofs): instruction slot `slot` of
// `n_slots` takes instructions begin + slot, + n_slots, ... of every level, a workgroup barrier ends a level (the waves of
// a workgroup share the CU's L1, so what one wave stored is what the next level's loads see).  A level costs about one
// memory round trip (the operands): reading the bounds and instruction words a level ahead was measured and gains nothing.
__device__ __forceinline__ void witness_walk(const WitnessArgs& a, const uint32_t* __restrict__ level_offsets, uint32_t l0, uint32_t l1,
                                             uint32_t p, uint32_t slot, uint32_t n_slots) {
    for (uint32_t l = l0; l < l1; l++) {
        const uint32_t begin = level_offsets[l], end = level_offsets[l + 1];
        if (p < a.n)
            for (uint32_t i = begin + slot; i < end; i += n_slots) witness_exec(a, i, p);
        __threadfence_block();
        __syncthreads();
    }
}

// The narrow tail of a program (the Poseidon AIR's accumulator chain: ~125 levels of 1 .. 28 instructions) in ONE launch:
// dependencies never cross proofs, so a workgroup that owns 64 proofs can walk the levels on its own — its four waves
// share a level's instructions, a workgroup barrier separates levels (the waves of a workgroup share the CU's L1, so what
// one wave stored is what the next level's loads see).
__global__ __launch_bounds__(256) void k_witness_strip(WitnessArgs a, const uint32_t* __restrict__ level_offsets, uint32_t l0, uint32_t l1) {
    const uint32_t p = blockIdx.x * 64 + (threadIdx.x & 63u), wv = threadIdx.x >> 6;
    witness_walk(a, level_offsets, l0, l1, p, wv, 4u);
}

// Small and medium batches: the WHOLE program in one launch.  A workgroup of 1 024 lanes owns P2 = 2^log_p2 proofs (n <= 4:
// one workgroup, P2 = n rounded up — the reference's own use is ONE proof per recursion step; larger batches: P2 proofs per
// workgroup, n / P2 workgroups side by side — dependencies never cross proofs) and walks the levels on its own:
// (1 024 / P2) instruction slots x P2 proofs, a level is a few rounds of slots, a workgroup barrier ends it.  265 barriers
// instead of 140 launches with a machine-wide drain between them; the price is narrower rows (P2 x 16 B per instruction
// instead of 1 KB), which is why the level-per-launch form takes over for large batches (witness_api.inc).
__global__ __launch_bounds__(1024) void k_witness_small(WitnessArgs a, const uint32_t* __restrict__ level_offsets, uint32_t n_levels,
                                                        uint32_t log_p2) {
    const uint32_t p = (blockIdx.x << log_p2) + (threadIdx.x & ((1u << log_p2) - 1u));
    witness_walk(a, level_offsets, 0, n_levels, p, threadIdx.x >> log_p2, 1024u >> log_p2);
}

// vars[variable][proof] -> out[proof][variable] (the reference's per-proof vector), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_witness_transpose(const uint4* __restrict__ vars, uint4* __restrict__ out, uint32_t n_vars,
                                                           uint32_t n) {
    __shared__ uint4 tile[32][33];
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    const uint32_t k0 = blockIdx.x * 32, p0 = blockIdx.y * 32;  // proofs (at most 2^20) on the y axis, whose limit is 65 535 blocks
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint32_t k = k0 + j, p = p0 + tx;
        if (k < n_vars && p < n) tile[j][tx] = vars[(size_t)k * n + p];
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint32_t p = p0 + j, k = k0 + tx;
        if (k < n_vars && p < n) out[(size_t)p * n_vars + k] = tile[tx][j];
    }
}

}  // namespace rsv
