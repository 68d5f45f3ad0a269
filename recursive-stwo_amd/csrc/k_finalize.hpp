// k_finalize.hpp — verdict and reason per proof (k_finalize), accept bitmap (k_bitmap).  Part of the pipeline described in verify.hpp.
#pragma once
#include "k_parse.hpp"

namespace rsv {

// --------------------------------------------------------------- k_finalize
// One lane per proof: first failing stage -> reason, accept byte; optionally the accept bitmap + count of the batch
// (rsv_hints_out::d_accept_bitmap: the buffer a multi-GPU host exchanges) in the same launch.  Before that the wave
// runs the canonicity fallback for those of its 64 proofs that a stage flagged F_RESCAN (layout.hpp: a witness list of
// the wrong length was not read completely): the wave reads such a proof once in full, the owning lane takes the
// result.  In a batch of well-formed and bit-flipped proofs nothing is flagged, and this is the only kernel behind the
// Merkle stages (rounds 1-2: a scan of every proof, 7.7 GB per 65 536-proof step; until the middle of round 3 a
// kernel of its own in front of this one).
// force: every parsed proof is read (the single-proof probe rsv_transcript, which runs no Merkle stage).
__global__ __launch_bounds__(256) void k_finalize(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets, uint32_t n,
                                                  const ProofMeta* __restrict__ metas, ProofCtx* __restrict__ ctxs,
                                                  uint8_t* __restrict__ accept, uint8_t* __restrict__ reason, uint32_t force,
                                                  uint32_t* __restrict__ bitmap, unsigned long long* __restrict__ count) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    // both words are fetched at once (k_parse has zeroed every proof's flags): a wave's 64 proofs sit in 64 different pages
    // of either array, and one such load after the other was most of this kernel's 34 us at 1 024 proofs
    uint32_t r = p < n ? metas[p].reason : (uint32_t)R_PARSE;
    uint32_t f = p < n ? ctxs[p].flags : 0u;
    if (r != R_OK) f = 0u;
    unsigned long long todo = __ballot(p < n && r == R_OK && (force || (f & F_RESCAN)));
    while (todo) {
        const uint32_t src = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1ull;
        const uint32_t q = p - lane + src;
        const uint32_t bad = scan_proof_words(blob, offsets, metas[q], q, lane);
        if (bad && lane == src) {
            f |= 1u << R_PARSE;
            atomicOr(&ctxs[q].flags, 1u << R_PARSE);  // k_export_transcript (behind this kernel) reads it
        }
    }
    if (p < n) {
        if (r == R_OK) {
            f &= F_REASON_MASK;
            r = f ? (uint32_t)(__ffs((int)f) - 1) : R_OK;
        }
        if (accept) accept[p] = r == R_OK;
        if (reason) reason[p] = (uint8_t)r;
    }
    if (bitmap) {
        const unsigned long long mask = __ballot(p < n && r == R_OK);
        if (lane == 0 && p < n) {
            bitmap[p >> 5] = (uint32_t)mask;
            if ((p >> 5) + 1 < (n + 31) / 32) bitmap[(p >> 5) + 1] = (uint32_t)(mask >> 32);
            if (count && mask) atomicAdd(count, (unsigned long long)__popcll(mask));
        }
    }
}

// accept bytes -> little-endian bitmap + popcount (the buffer the multi-GPU host all-gathers)
__global__ __launch_bounds__(256) void k_bitmap(const uint8_t* __restrict__ accept, uint32_t n,
                                                uint32_t* __restrict__ bitmap, unsigned long long* __restrict__ count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = i < n && accept[i];
    unsigned long long mask = __ballot(bit);
    uint32_t lane = threadIdx.x & 63;
    if (lane == 0 && i < n) {
        bitmap[i >> 5] = (uint32_t)mask;
        if ((i >> 5) + 1 < (n + 31) / 32) bitmap[(i >> 5) + 1] = (uint32_t)(mask >> 32);
        if (count && mask) atomicAdd(count, (unsigned long long)__popcll(mask));
    }
}

}  // namespace rsv
