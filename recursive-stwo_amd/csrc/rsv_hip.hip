// rsv_hip.hip — the C-ABI of include/rsv.h implemented on HIP for gfx950.
//
// This translation unit is the whole product library (librsv_hip.so).  It
// contains no CPU verification path: every entry point either runs the HIP
// kernels or fails with RSV_E_DEVICE.
#include "../../include/rsv.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "primitives.hpp"
#include "k_emulated.hpp"
#include "verify.hpp"
#include "host_logic.hpp"
#include "k_witness.hpp"

using namespace rsv;

// Options (include/rsv.h: rsv_option).  Nothing in this library reads the environment: a knob is set through
// rsv_ctx_set_option, on one context or (ctx == NULL) as the process default that contexts created later inherit.
struct Options {
    int transcript_form = 0;   // 0 auto (by batch size), 1 row, 2 lane
    int transcript_split = 0;  // 0 auto, 1 one launch, 2 front + back
    int oods_form = 0, qconst_form = 0;  // 0 auto, 1 row, 2 lane
    int plan_form = 0;         // 0 / 1 one lane per (proof, query), 2 one lane per proof (the original)
    int tree_cap = 0;          // 0 / 1 dense top-of-tree cap, 2 every path walks to the root
    int overlap_trees = 0;     // 0 / 1 FRI trees beside the trace trees, 2 behind them
    long long ws_budget_mb = 8192;
    int perm_wg_per_cu = 24;
    int perm_form = 0;         // experiment: instance / launch bound of k_permute (primitives.hpp)
    long long host_chunk_mb = 256;
    int host_threads = 0;      // 0 = min(cores, 4)
    int critical_chain = 0;    // 0 auto, 1 the chain of dependent kernels on ONE stream, 2 the round-2 stream layout
    int device_order = 0;      // 0 / 1 single-configuration batches: slot order on the device, no host round trip; 2 host
    int graph = 0;             // 0 / 2 off, 1 replay repeated identical calls as a HIP graph (experiment)
    int witness_layout = 0;    // 0 / 1 d_variables [proof][variable], 2 [variable][proof]
    int flow_cap = 0;          // PoseidonFlow passes: 0 / 1 top-of-tree cap (shared nodes hashed once), 2 every lane walks to the root
    int query_form = 0;        // 0 auto (by batch size), 1 k_query with a row of 16 threads per query, 2 with one lane per query
    int stage_times = 0;       // 0 / 2 off, 1 record an event pair around every stage (rsv_last_stage_times)
    int tree_pace = 0;         // 0 auto (by batch size), 1 the tree kernels on the paced permutation instances, 2 on the unpaced ones, 3 in the row form (16 threads per path)
    int pair_order = 0;        // 0 / 1 the FRI trees of a small launch dealt out over the compute units, 2 grid row y = tree y
    int cap_mid = 0;           // 0 auto (buckets whose in-kernel cap levels fill their waves badly), 1 every bucket hands over at the cap level (k_cap_mid + k_cap_top), 2 none
    int oods_early = 0;        // 0 / 2 the OODS check behind the trace trees (side stream), 1 (experiment) right behind the transcript on the aux stream
    int tree_order = 0;        // 0 / 2 the tree kernels' grid row y = tree, 1 (measured, slower) their workgroups in the XCD-aware interleaved order
    int cap_top = 0;           // 0 auto (batches of >= 1 024 proofs), 1 the last levels of every tree in k_cap_top, 2 inside the Merkle kernels
    long long witness_small_max = 0;  // 0 default, else 1 + the largest batch that runs the program in one launch
    int witness_small_log = 0;        // 0 default, else 1 + log2(proofs per workgroup) of that form
    int witness_walk_log = 0;         // level form: 0 auto, 1 off, else 1 + log2(proofs per workgroup) of the one-launch tail
};
static Options g_default_options;
static std::mutex g_options_mu;
static std::atomic<int> g_debug_log{0};

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            if (g_debug_log.load(std::memory_order_relaxed))                                       \
                fprintf(stderr, "rsv: %s -> %s\n", #expr, hipGetErrorString(e_));                  \
            return RSV_E_DEVICE;                                                                   \
        }                                                                                          \
    } while (0)

namespace {
struct VerifyState;                       // stage clock of the last verify call (verify_api.inc)
void destroy_verify_state(VerifyState*);
struct HostPipe;                          // pinned staging ring of rsv_verify_batch_host (host_stream.inc)
void destroy_host_pipe(HostPipe*);
}  // namespace
struct GraphCache;                        // RSV_OPT_GRAPH (verify_api.inc)
static void destroy_graph_cache(GraphCache*);

struct rsv_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;  // row hashes, quotient constants, k_query, FRI trees: underneath the main stream
    hipStream_t aux = nullptr;   // the front half of a small batch's transcript, next to the parser
    hipEvent_t ev_begin = nullptr, ev_front = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_scan = nullptr, ev_plan = nullptr, ev_query = nullptr, ev_tr = nullptr, ev_ids = nullptr, ev_ext = nullptr, ev_oods = nullptr;
    // reusable HBM workspace for rsv_verify_batch_dev
    void* ws = nullptr;        // per-query stages (plan, FRI leaf values)
    size_t ws_bytes = 0;
    void* ws_fixed = nullptr;  // per-proof records of the current batch
    size_t ws_fixed_bytes = 0;
    void* ws_rows = nullptr;   // k_row_hash output
    size_t ws_rows_bytes = 0;
    void* ws_witness = nullptr;  // rsv_witness_eval_dev: the hints it asks the verifying pass for, and variables[var][proof]
    size_t ws_witness_bytes = 0;
    const struct rsv::ProofMeta* last_metas = nullptr;  // the parser's records of the last batch (inside ws_fixed)
    VerifyState* vs = nullptr;
    rsv_public_input* d_pi = nullptr;
    size_t d_pi_cap = 0;
    HostPipe* host_pipe = nullptr;
    GraphCache* graphs = nullptr;
    Options opt;
};

namespace {

// RAII device buffer for the host-pointer convenience entry points.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RSV_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RSV_E_DEVICE;
    return RSV_OK;
}

Options default_options() {
    std::lock_guard<std::mutex> lk(g_options_mu);
    return g_default_options;
}

inline unsigned grid_for(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// k_permute walks the states with a grid stride from a grid sized to the machine, not to n: RSV_OPT_PERM_WG_PER_CU workgroups
// of 256 lanes per CU (8 are resident at the kernel's 64 registers).  Measured on 2^24 states (tools/perm_bench.py, round 5:
// gpurun_out/r5_h/perm.txt): 8.85 / 9.20 / 9.38 / 9.46 / 9.58 / 9.75 G permutations/s at 6 / 8 / 10 / 12 / 16 / 24 — with
// exactly the resident number every lane walks the same count and the launch ends on its slowest wave; three times as many,
// a third as long each, even that out (default 24; until round 4: 8, 8.83 G/s on that round's boxes).  Never more than one lane per state.
inline unsigned permute_grid(size_t n, int wg_per_cu) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned per_cu = (unsigned)wg_per_cu;
    const size_t want = (size_t)cus * per_cu;
    const size_t need = (n + 255) / 256;
    return (unsigned)(need < want ? need : want);
}

}  // namespace

extern "C" {

int rsv_abi_version(void) { return RSV_ABI_VERSION; }

int rsv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rsv_ctx_create(int device, rsv_ctx** out) {
    if (!out) return RSV_E_NULL;
    *out = nullptr;
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    rsv_ctx* c = new (std::nothrow) rsv_ctx();
    if (!c) return RSV_E_DEVICE;
    c->device = device;
    c->opt = default_options();
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu < 1) c->n_cu = 256;
    // The side stream carries the kernels another stage waits for while they share the machine with a wider one (k_query
    // beside the trace trees): one dispatch priority level above normal where the device has one, a plain stream otherwise.
    auto create_side = [](hipStream_t* s) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest <= -1 &&
            hipStreamCreateWithPriority(s, hipStreamNonBlocking, -1) == hipSuccess)
            return hipSuccess;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        create_side(&c->side) != hipSuccess ||
        hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_scan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tr, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ids, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ext, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_oods, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_query, hipEventDisableTiming) != hipSuccess) {
        rsv_ctx_destroy(c);
        return RSV_E_DEVICE;
    }
    *out = c;
    return RSV_OK;
}

void rsv_ctx_destroy(rsv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_scan) (void)hipEventDestroy(c->ev_scan);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    if (c->ev_tr) (void)hipEventDestroy(c->ev_tr);
    if (c->ev_ids) (void)hipEventDestroy(c->ev_ids);
    if (c->ev_query) (void)hipEventDestroy(c->ev_query);
    if (c->ev_ext) (void)hipEventDestroy(c->ev_ext);
    if (c->ev_oods) (void)hipEventDestroy(c->ev_oods);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->vs) destroy_verify_state(c->vs);
    if (c->host_pipe) destroy_host_pipe(c->host_pipe);
    if (c->graphs) destroy_graph_cache(c->graphs);
    if (c->ws) (void)hipFree(c->ws);
    if (c->ws_fixed) (void)hipFree(c->ws_fixed);
    if (c->ws_rows) (void)hipFree(c->ws_rows);
    if (c->ws_witness) (void)hipFree(c->ws_witness);
    if (c->d_pi) (void)hipFree(c->d_pi);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rsv_ctx_synchronize(rsv_ctx* c) {
    if (!c) return RSV_E_NULL;
    // every verify call joins its side streams into the main one before it returns — except on an error exit taken
    // after the fork: wait for all three, so that a caller who synchronises after an error may free its buffers
    HIP_TRY(hipStreamSynchronize(c->side));
    HIP_TRY(hipStreamSynchronize(c->aux));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSV_OK;
}

void* rsv_ctx_stream(rsv_ctx* c) { return c ? (void*)c->stream : nullptr; }

int rsv_ctx_set_option(rsv_ctx* c, int option, long long value) {
    auto tri = [&](int* dst) { if (value < 0 || value > 2) return (int)RSV_E_RANGE; *dst = (int)value; return (int)RSV_OK; };
    std::lock_guard<std::mutex> lk(g_options_mu);
    Options& o = c ? c->opt : g_default_options;
    switch (option) {
        case RSV_OPT_TRANSCRIPT_FORM: return tri(&o.transcript_form);
        case RSV_OPT_TRANSCRIPT_SPLIT: return tri(&o.transcript_split);
        case RSV_OPT_OODS_FORM: return tri(&o.oods_form);
        case RSV_OPT_QCONST_FORM: return tri(&o.qconst_form);
        case RSV_OPT_PLAN_FORM: return tri(&o.plan_form);
        case RSV_OPT_TREE_CAP: return tri(&o.tree_cap);
        case RSV_OPT_OVERLAP_TREES: return tri(&o.overlap_trees);
        case RSV_OPT_CRITICAL_CHAIN: return tri(&o.critical_chain);
        case RSV_OPT_DEVICE_ORDER: return tri(&o.device_order);
        case RSV_OPT_GRAPH: return tri(&o.graph);
        case RSV_OPT_WITNESS_LAYOUT: return tri(&o.witness_layout);
        case RSV_OPT_CAP_TOP: return tri(&o.cap_top);
        case RSV_OPT_FLOW_CAP: return tri(&o.flow_cap);
        case RSV_OPT_PAIR_ORDER: return tri(&o.pair_order);
        case RSV_OPT_TREE_PACE: if (value < 0 || value > 3) return (int)RSV_E_RANGE; o.tree_pace = (int)value; return (int)RSV_OK;
        case RSV_OPT_STAGE_TIMES: return tri(&o.stage_times);
        case RSV_OPT_QUERY_FORM: return tri(&o.query_form);
        case RSV_OPT_CAP_MID: return tri(&o.cap_mid);
        case RSV_OPT_TREE_ORDER: return tri(&o.tree_order);
        case RSV_OPT_OODS_EARLY: return tri(&o.oods_early);
        case RSV_OPT_WITNESS_WALK_LOG:
            if (value < 0 || value > 7) return RSV_E_RANGE;
            o.witness_walk_log = (int)value; return RSV_OK;
        case RSV_OPT_WITNESS_SMALL_MAX:
            if (value < 0 || value > (1ll << 20) + 1) return RSV_E_RANGE;
            o.witness_small_max = value; return RSV_OK;
        case RSV_OPT_WITNESS_SMALL_LOG:
            if (value < 0 || value > 7) return RSV_E_RANGE;
            o.witness_small_log = (int)value; return RSV_OK;
        case RSV_OPT_WS_BUDGET_MB:
            if (value < 1 || value > (1ll << 20)) return RSV_E_RANGE;
            o.ws_budget_mb = value; return RSV_OK;
        case RSV_OPT_PERM_FORM:
            if (value < 0 || value > 2) return RSV_E_RANGE;
            o.perm_form = (int)value; return RSV_OK;
        case RSV_OPT_PERM_WG_PER_CU:
            if (value < 1 || value > 32) return RSV_E_RANGE;
            o.perm_wg_per_cu = (int)value; return RSV_OK;
        case RSV_OPT_HOST_CHUNK_MB:
            if (value < 1 || value > 16384) return RSV_E_RANGE;
            o.host_chunk_mb = value; return RSV_OK;
        case RSV_OPT_HOST_THREADS:
            if (value < 0 || value > 64) return RSV_E_RANGE;
            o.host_threads = (int)value; return RSV_OK;
        case RSV_OPT_DEBUG_LOG:
            if (value < 0 || value > 1) return RSV_E_RANGE;
            g_debug_log.store((int)value); return RSV_OK;
        default: return RSV_E_SIZE;
    }
}

// Ordering against the caller's own streams.  Every entry point forks its side stream off the main one (ev_fork), so
// ordering the main stream is enough.
int rsv_ctx_wait_stream(rsv_ctx* c, void* hip_stream) {
    if (!c) return RSV_E_NULL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev_ext, static_cast<hipStream_t>(hip_stream)));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ext, 0));
    return RSV_OK;
}
int rsv_stream_wait_ctx(rsv_ctx* c, void* hip_stream) {
    if (!c) return RSV_E_NULL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev_ext, c->stream));
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), c->ev_ext, 0));
    return RSV_OK;
}

// ---------------------------------------------------------------- a3
int rsv_poseidon2_permute_dev(rsv_ctx* c, const uint32_t* d_in, uint32_t* d_out, size_t n, uint32_t* d_bad) {
    if (!c || (n && (!d_in || !d_out))) return RSV_E_NULL;
    if (n == 0) return RSV_OK;
    if (n > ((size_t)1 << 31)) return RSV_E_SIZE;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return RSV_E_SIZE;
    HIP_TRY(hipSetDevice(c->device));
    {
        const dim3 grid(permute_grid(n, c->opt.perm_wg_per_cu));
        const uint4* pin = reinterpret_cast<const uint4*>(d_in);
        uint4* pout = reinterpret_cast<uint4*>(d_out);
        switch (c->opt.perm_form) {  // RSV_OPT_PERM_FORM: an experiment's knob
            case 1: hipLaunchKernelGGL((k_permute<1, 1>), grid, dim3(256), 0, c->stream, pin, pout, n, d_bad); break;
            case 2: hipLaunchKernelGGL((k_permute<2, 1>), grid, dim3(256), 0, c->stream, pin, pout, n, d_bad); break;
            default: hipLaunchKernelGGL((k_permute<0, 1>), grid, dim3(256), 0, c->stream, pin, pout, n, d_bad); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return RSV_OK;
}

int rsv_poseidon2_permute(const uint32_t* in16, uint32_t* out16, size_t n, int device) {
    if (n && (!in16 || !out16)) return RSV_E_NULL;
    if (n > ((size_t)1 << 28)) return RSV_E_SIZE;
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    if (n == 0) return RSV_OK;
    DevBuf din, dout, dbad;
    HIP_TRY(din.alloc(64 * n));
    HIP_TRY(dout.alloc(64 * n));
    HIP_TRY(dbad.alloc(4));
    HIP_TRY(hipMemset(dbad.p, 0, 4));
    HIP_TRY(hipMemcpy(din.p, in16, 64 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_permute<0, 1>), dim3(permute_grid(n, default_options().perm_wg_per_cu)), dim3(256), 0, 0, din.as<const uint4>(),
                       dout.as<uint4>(), n, dbad.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return RSV_E_RANGE;
    HIP_TRY(hipMemcpy(out16, dout.p, 64 * n, hipMemcpyDeviceToHost));
    return RSV_OK;
}

// ---------------------------------------------------------------- a4
int rsv_poseidon2_half_permute(const uint32_t* left8, const uint32_t* right8, const uint8_t* swap,
                               uint32_t* out_rate8, uint32_t* out_cap8, size_t n, int device) {
    if (n && (!left8 || !right8)) return RSV_E_NULL;
    if (n > ((size_t)1 << 28)) return RSV_E_SIZE;
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    if (n == 0) return RSV_OK;
    DevBuf dl, dr, ds, drate, dcap, dbad;
    HIP_TRY(dl.alloc(32 * n));
    HIP_TRY(dr.alloc(32 * n));
    HIP_TRY(dbad.alloc(4));
    HIP_TRY(hipMemset(dbad.p, 0, 4));
    HIP_TRY(hipMemcpy(dl.p, left8, 32 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, right8, 32 * n, hipMemcpyHostToDevice));
    if (swap) {
        HIP_TRY(ds.alloc(n));
        HIP_TRY(hipMemcpy(ds.p, swap, n, hipMemcpyHostToDevice));
    }
    if (out_rate8) HIP_TRY(drate.alloc(32 * n));
    if (out_cap8) HIP_TRY(dcap.alloc(32 * n));
    hipLaunchKernelGGL(k_half_permute, dim3(grid_for(n, 256)), dim3(256), 0, 0, dl.as<const uint32_t>(),
                       dr.as<const uint32_t>(), swap ? ds.as<const uint8_t>() : nullptr,
                       out_rate8 ? drate.as<uint32_t>() : nullptr, out_cap8 ? dcap.as<uint32_t>() : nullptr, n,
                       dbad.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return RSV_E_RANGE;
    if (out_rate8) HIP_TRY(hipMemcpy(out_rate8, drate.p, 32 * n, hipMemcpyDeviceToHost));
    if (out_cap8) HIP_TRY(hipMemcpy(out_cap8, dcap.p, 32 * n, hipMemcpyDeviceToHost));
    return RSV_OK;
}

// ---------------------------------------------------------------- f4
static_assert(RSV_EMU_STRIDE == EMU_STRIDE && RSV_EMU_ROWS == EMU_ROWS && RSV_EMU_SWAP_ROWS == EMU_SWAP_ROWS,
              "rsv.h and k_emulated.hpp disagree");

int rsv_poseidon2_emulated_dev(rsv_ctx* c, const uint32_t* d_left, const uint32_t* d_right, const uint8_t* d_swap,
                               uint32_t* d_rows, size_t n, uint32_t* d_bad) {
    if (!c || (n && (!d_left || !d_right || !d_rows))) return RSV_E_NULL;
    if (n == 0) return RSV_OK;
    if (n > ((size_t)1 << 26)) return RSV_E_SIZE;
    if (((uintptr_t)d_left & 15) || ((uintptr_t)d_right & 15) || ((uintptr_t)d_rows & 15)) return RSV_E_SIZE;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_emulated, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, d_left, d_right, d_swap,
                       reinterpret_cast<uint4*>(d_rows), n, d_bad);
    HIP_TRY(hipGetLastError());
    return RSV_OK;
}

int rsv_poseidon2_emulated(const uint32_t* left8, const uint32_t* right8, const uint8_t* swap, uint32_t* rows, size_t n,
                           int device) {
    if (n && (!left8 || !right8 || !rows)) return RSV_E_NULL;
    if (n > ((size_t)1 << 22)) return RSV_E_SIZE;  // 2^22 permutations = 27.9 GB of rows
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    if (n == 0) return RSV_OK;
    const size_t row_bytes = (size_t)EMU_STRIDE * 16 * n;
    DevBuf dl, dr, ds, drows, dbad;
    HIP_TRY(dl.alloc(32 * n));
    HIP_TRY(dr.alloc(32 * n));
    HIP_TRY(drows.alloc(row_bytes));
    HIP_TRY(dbad.alloc(4));
    HIP_TRY(hipMemset(dbad.p, 0, 4));
    HIP_TRY(hipMemcpy(dl.p, left8, 32 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, right8, 32 * n, hipMemcpyHostToDevice));
    if (swap) {
        HIP_TRY(ds.alloc(n));
        HIP_TRY(hipMemcpy(ds.p, swap, n, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_emulated, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, dl.as<const uint32_t>(),
                       dr.as<const uint32_t>(), swap ? ds.as<const uint8_t>() : nullptr, drows.as<uint4>(), n,
                       dbad.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return RSV_E_RANGE;
    HIP_TRY(hipMemcpy(rows, drows.p, row_bytes, hipMemcpyDeviceToHost));
    return RSV_OK;
}

// ---------------------------------------------------------------- a5
int rsv_merkle_hash_node(const uint32_t* left8, const uint32_t* right8, const uint32_t* cols, size_t n_cols,
                         uint32_t* out8, size_t n, int device) {
    if (n && !out8) return RSV_E_NULL;
    if ((left8 == nullptr) != (right8 == nullptr)) return RSV_E_NULL;
    if (n_cols && !cols) return RSV_E_NULL;
    if (!left8 && n_cols == 0) return RSV_E_SIZE;
    if (n > ((size_t)1 << 26) || n_cols > 4096) return RSV_E_SIZE;
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    if (n == 0) return RSV_OK;
    DevBuf dl, dr, dc, dout, dbad;
    HIP_TRY(dbad.alloc(4));
    HIP_TRY(hipMemset(dbad.p, 0, 4));
    if (left8) {
        HIP_TRY(dl.alloc(32 * n));
        HIP_TRY(dr.alloc(32 * n));
        HIP_TRY(hipMemcpy(dl.p, left8, 32 * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dr.p, right8, 32 * n, hipMemcpyHostToDevice));
    }
    if (n_cols) {
        HIP_TRY(dc.alloc(4 * n_cols * n));
        HIP_TRY(hipMemcpy(dc.p, cols, 4 * n_cols * n, hipMemcpyHostToDevice));
    }
    HIP_TRY(dout.alloc(32 * n));
    hipLaunchKernelGGL(k_hash_node, dim3(grid_for(n, 256)), dim3(256), 0, 0,
                       left8 ? dl.as<const uint32_t>() : nullptr, left8 ? dr.as<const uint32_t>() : nullptr,
                       n_cols ? dc.as<const uint32_t>() : nullptr, (uint32_t)n_cols, dout.as<uint32_t>(), n,
                       dbad.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, dbad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return RSV_E_RANGE;
    HIP_TRY(hipMemcpy(out8, dout.p, 32 * n, hipMemcpyDeviceToHost));
    return RSV_OK;
}

// ---------------------------------------------------------------- a9
int rsv_merkle_path_root(const uint32_t* query, const uint32_t* sib8, const uint32_t* cols,
                         const uint32_t* n_cols_at, uint32_t depth, uint32_t* out_root8, size_t n, int device) {
    if (!query || !n_cols_at || !out_root8 || !cols || (depth && !sib8)) return RSV_E_NULL;
    if (depth > 31 || n_cols_at[depth] == 0 || n > ((size_t)1 << 24)) return RSV_E_SIZE;
    size_t per_path = 0;
    for (uint32_t h = 0; h <= depth; h++) {
        if (n_cols_at[h] > 4096) return RSV_E_SIZE;
        per_path += n_cols_at[h];
    }
    int rc = select_device(device);
    if (rc != RSV_OK) return rc;
    if (n == 0) return RSV_OK;
    DevBuf dq, ds, dc, dn, dout;
    HIP_TRY(dq.alloc(4 * n));
    HIP_TRY(ds.alloc(32 * (size_t)depth * n));
    HIP_TRY(dc.alloc(4 * per_path * n));
    HIP_TRY(dn.alloc(4 * (depth + 1)));
    HIP_TRY(dout.alloc(32 * n));
    HIP_TRY(hipMemcpy(dq.p, query, 4 * n, hipMemcpyHostToDevice));
    if (depth) HIP_TRY(hipMemcpy(ds.p, sib8, 32 * (size_t)depth * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.p, cols, 4 * per_path * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dn.p, n_cols_at, 4 * (depth + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_path_root, dim3(grid_for(n, 256)), dim3(256), 0, 0, dq.as<const uint32_t>(),
                       ds.as<const uint32_t>(), dc.as<const uint32_t>(), dn.as<const uint32_t>(), depth,
                       (uint32_t)per_path, dout.as<uint32_t>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out_root8, dout.p, 32 * n, hipMemcpyDeviceToHost));
    return RSV_OK;
}

}  // extern "C"

#include "verify_api.inc"
#include "host_stream.inc"
#include "multi_api.inc"
#include "circuit_program.hpp"
#include "witness_api.inc"
#include "circuit_builder.inc"
