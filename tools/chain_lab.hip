// chain_lab.hip — what does a boundary between two DEPENDENT stages cost on MI355X, and what would ONE persistent launch
// with grid-wide barriers cost instead?  ("neither streams nor graphs remove the 12-25 us per boundary".)
//
// A small batch's verify call is a chain of ~8 dependent kernels.  This lab runs the same synthetic chain three ways:
//   launches    K kernels on one stream, each reading what the previous one wrote (the product's chain layout)
//   graph       the same K launches captured once and replayed as a HIP graph
//   persistent  ONE launch of a grid that is resident all at once (<= n_cu x 4 workgroups); between stages every workgroup
//               arrives at a device-scope atomic counter and spins until all have (the only grid barrier there is for
//               code that is not a cooperative launch), with the fences a consumer of the previous stage's data needs
// Each stage does `work` dependent multiply-adds per lane (0 = empty stage: pure boundary cost) and passes one word per
// lane through global memory.  Output: microseconds per stage for each form and grid size.
// Build: hipcc --offload-arch=gfx950 -O3 -o chain_lab chain_lab.hip ; run: ./chain_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(e)                                                                       \
    do {                                                                               \
        hipError_t e_ = (e);                                                           \
        if (e_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } \
    } while (0)

__device__ __forceinline__ uint32_t stage_work(uint32_t x, int work) {
    for (int i = 0; i < work; i++) x = x * 2654435761u + 12345u;
    return x;
}

__global__ __launch_bounds__(256) void k_stage(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int work, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // read a neighbour's word: a real cross-lane dependency on the previous stage
    out[i] = stage_work(in[(i + 257u) % n], work);
}

// the same stage with what the product's launches carry: a 2 KB by-value argument (Fused<..Args>), scratch memory, or both
struct BigArg { uint32_t pad[500]; const uint32_t* in; uint32_t* out; int work; uint32_t n; };
template <bool SCRATCH>
__global__ __launch_bounds__(256) void k_stage_big(BigArg a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = a.in[(i + 257u) % a.n] + a.pad[i % 500u];
    if (SCRATCH) {
        volatile uint32_t spill[64];  // indexed at run time: lives in scratch
        for (int k = 0; k < 64; k++) spill[k] = v + k;
        v = spill[(v >> 3) & 63u];
    }
    a.out[i] = stage_work(v, a.work);
}

// all workgroups of the grid are resident (the host sizes the grid for that): arrive + spin, generation by generation
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's stores before the arrival
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_persistent(uint32_t* a, uint32_t* b, int stages, int work, uint32_t n, unsigned* counter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *in = a, *out = b;
    for (int s = 0; s < stages; s++) {
        // the previous stage's words were written by other workgroups: read them past the vector L1
        const uint32_t v = __hip_atomic_load(&in[(i + 257u) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[i] = stage_work(v, work);
        if (s + 1 < stages) grid_barrier(counter, (unsigned)(s + 1) * gridDim.x);
        uint32_t* t = in; in = out; out = t;
    }
}

int main() {
    int n_cu = 0;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int stages = 8, reps = 200;
    printf("%d compute units; %d dependent stages; us per stage (boundary included)\n", n_cu, stages);
    printf("%8s %6s %10s %10s %12s %10s %10s %10s\n", "blocks", "work", "launches", "graph", "persistent", "+events", "+2KB arg", "+scratch");
    std::vector<hipEvent_t> ev(2 * stages);
    for (auto& e : ev) CHECK(hipEventCreate(&e));
    for (int blocks : {1, 64, 256, 1024}) {
        if (blocks > n_cu * 4) continue;
        const uint32_t n = (uint32_t)blocks * 256u;
        uint32_t *a, *b;
        unsigned* counter;
        CHECK(hipMalloc(&a, 4 * (size_t)n));
        CHECK(hipMalloc(&b, 4 * (size_t)n));
        CHECK(hipMalloc(&counter, 4));
        CHECK(hipMemset(a, 1, 4 * (size_t)n));
        for (int work : {0, 2000, 20000}) {
            auto chain = [&]() {
                uint32_t *in = a, *out = b;
                for (int s = 0; s < stages; s++) {
                    hipLaunchKernelGGL(k_stage, dim3(blocks), dim3(256), 0, st, in, out, work, n);
                    uint32_t* t = in; in = out; out = t;
                }
            };
            auto time_us = [&](auto&& body) {
                for (int w = 0; w < 5; w++) body();
                (void)hipStreamSynchronize(st);
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; r++) { body(); (void)hipStreamSynchronize(st); }  // one call at a time, like a verify call
                const auto t1 = std::chrono::steady_clock::now();
                return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / stages;
            };
            const double t_launch = time_us(chain);
            hipGraph_t graph;
            hipGraphExec_t exec;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            chain();
            CHECK(hipStreamEndCapture(st, &graph));
            CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            const double t_graph = time_us([&]() { (void)hipGraphLaunch(exec, st); });
            const double t_pers = time_us([&]() {
                (void)hipMemsetAsync(counter, 0, 4, st);
                hipLaunchKernelGGL(k_persistent, dim3(blocks), dim3(256), 0, st, a, b, stages, work, n, counter);
            });
            // the product's stage clock: an event before and behind every stage
            const double t_events = time_us([&]() {
                uint32_t *in = a, *out = b;
                for (int s = 0; s < stages; s++) {
                    (void)hipEventRecord(ev[2 * s], st);
                    hipLaunchKernelGGL(k_stage, dim3(blocks), dim3(256), 0, st, in, out, work, n);
                    (void)hipEventRecord(ev[2 * s + 1], st);
                    uint32_t* t = in; in = out; out = t;
                }
            });
            auto big_chain = [&](bool scratch) {
                BigArg arg{};
                arg.in = a; arg.out = b; arg.work = work; arg.n = n;
                for (int s = 0; s < stages; s++) {
                    if (scratch) hipLaunchKernelGGL(k_stage_big<true>, dim3(blocks), dim3(256), 0, st, arg);
                    else hipLaunchKernelGGL(k_stage_big<false>, dim3(blocks), dim3(256), 0, st, arg);
                    const uint32_t* t = arg.in; arg.in = arg.out; arg.out = const_cast<uint32_t*>(t);
                }
            };
            const double t_big = time_us([&]() { big_chain(false); });
            const double t_scr = time_us([&]() { big_chain(true); });
            printf("%8d %6d %10.2f %10.2f %12.2f %10.2f %10.2f %10.2f\n", blocks, work, t_launch, t_graph, t_pers, t_events, t_big, t_scr);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
        }
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(counter);
    }
    return 0;
}
