// k_finalize.hpp — verdict and reason per proof (k_finalize), accept bitmap (k_bitmap).  Part of the pipeline described in verify.hpp.
#pragma once
#include "verify_common.hpp"

namespace rsv {

// --------------------------------------------------------------- k_finalize
__global__ __launch_bounds__(256) void k_finalize(uint32_t n, const ProofMeta* __restrict__ metas,
                                                  const ProofCtx* __restrict__ ctxs, uint8_t* __restrict__ accept,
                                                  uint8_t* __restrict__ reason) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t r = metas[p].reason;
    if (r == R_OK) {
        uint32_t f = ctxs[p].flags & F_REASON_MASK;
        r = f ? (uint32_t)(__ffs((int)f) - 1) : R_OK;
    }
    accept[p] = r == R_OK;
    if (reason) reason[p] = (uint8_t)r;
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
