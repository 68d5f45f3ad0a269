// verify_common.hpp — constants and small helpers shared by the verify kernels.  Part of the pipeline described in verify.hpp.
#pragma once
#include "circle.hpp"
#include "layout.hpp"
#include "merkle.hpp"
#include "poseidon2_row.hpp"

namespace rsv {

__constant__ SampleTable SAMPLES = make_sample_table();

enum : uint32_t {
    R_OK = 0, R_PARSE = 1, R_POW = 2, R_LOGUP = 3, R_COMPOSITION = 4, R_DUP_QUERY = 5, R_MERKLE_T0 = 6,
    R_FRI_FIRST = 10, R_FRI_INNER = 11, R_FRI_LAST = 12
};

struct PubInput {
    uint32_t idx;
    uint32_t value[4];
};

// The PCS configurations the caller expects (rsv_cfg_set, include/rsv.h): a proof must carry exactly the configuration
// its cfg_of entry names (entry 0 when cfg_of is null), else RSV_R_PARSE.  The reference never reads the configuration
// from the proof (FiatShamirHints::new(&proof, config, ..), components/hints/src/fiat_shamir.rs:69-74).
constexpr int MAX_CFGS = 16;
struct CfgSet {
    uint32_t n;
    uint32_t c[MAX_CFGS][4];  // pow_bits, log_blowup_factor, log_last_layer_degree_bound, n_queries
    const uint8_t* cfg_of;    // device pointer, one index per proof, or null
};

__device__ __forceinline__ QM31 ldq(const uint32_t* p) { return q_mk(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void stq(uint32_t* p, QM31 v) { p[0] = v.a.a; p[1] = v.a.b; p[2] = v.b.a; p[3] = v.b.b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

}  // namespace rsv
