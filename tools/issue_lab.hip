// issue_lab.hip — how do "fast" (v_add_u32 class) and "normal" (v_mul_lo_u32 class) VALU ops share a SIMD
// on gfx950?  Mixes of the two inside one wave, and waves that run only one class each.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_lab issue_lab.hip ; run: ./issue_lab [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define ITERS 32768
#define CH 8

#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b))
#define MUL(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m[i]) : "v"(b))
#define MAD(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(m[i]), "v"(b) : "s10", "s11")
#define SHR(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]))

#define PROLOG                                                                                   \
    uint32_t a[CH], m[CH], b = seed | 1u;                                                        \
    uint64_t w[CH];                                                                              \
    for (int i = 0; i < CH; i++) { a[i] = threadIdx.x * 2654435761u + i + seed; m[i] = a[i] ^ 77u; w[i] = a[i]; }
#define EPILOG                                                                                   \
    uint32_t r = 0;                                                                              \
    for (int i = 0; i < CH; i++) r ^= a[i] ^ m[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);     \
    if (r == 0x12345678u) out[0] = r;

// n_add adds and n_mul muls per chain step, interleaved
template <int NA, int NM, int KIND>
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, uint32_t seed) {
    PROLOG
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (KIND == 0) {
                if (NA > 0) ADD(i);
                if (NM > 0) MUL(i);
                if (NA > 1) ADD(i);
                if (NM > 1) MUL(i);
                if (NA > 2) ADD(i);
                if (NA > 3) ADD(i);
            } else {
                if (NA > 0) SHR(i);
                if (NM > 0) MAD(i);
                if (NA > 1) ADD(i);
                if (NM > 1) MAD(i);
                if (NA > 2) ADD(i);
                if (NA > 3) SHR(i);
            }
        }
    }
    EPILOG
}
// waves alternate: even waves only add (2*CH*... same instruction count), odd waves only mul
__global__ __launch_bounds__(256) void k_split(uint32_t* out, uint32_t seed, int mode) {
    PROLOG
    const int wave = (threadIdx.x >> 6) + (blockIdx.x & 1);  // neighbouring blocks flip, so each SIMD sees both kinds
    const bool adder = mode == 0 ? (wave & 1) : mode == 1;
    if (adder) {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < CH; i++) { ADD(i); ADD(i); }
        }
    } else {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < CH; i++) { MUL(i); MUL(i); }
        }
    }
    EPILOG
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    uint32_t* out; hipMalloc(&out, 4096);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int blocks = prop.multiProcessorCount * wps;
    double steps = (double)wps * ITERS * CH;  // chain steps per SIMD
    auto rep = [&](const char* name, float ms, int n_inst) {
        double ns = ms * 1e6 / steps;
        printf("%-34s %8.3f ms  %6.2f cyc/step  %5.2f cyc/inst\n", name, ms, ns * 2.4, ns * 2.4 / n_inst);
    };
#define RUN(NA, NM, KIND) rep(#NA " fast + " #NM " normal, kind " #KIND, time_it([&] { hipLaunchKernelGGL((k_mix<NA, NM, KIND>), dim3(blocks), dim3(256), 0, 0, out, 1u); }), NA + NM)
    printf("waves/SIMD %d\n", wps);
    RUN(1, 0, 0); RUN(0, 1, 0); RUN(1, 1, 0); RUN(2, 1, 0); RUN(3, 1, 0); RUN(4, 1, 0); RUN(2, 2, 0); RUN(4, 2, 0);
    RUN(1, 0, 1); RUN(0, 1, 1); RUN(1, 1, 1); RUN(2, 1, 1); RUN(3, 1, 1); RUN(4, 1, 1); RUN(4, 2, 1);
    rep("split waves: add-only | mul-only", time_it([&] { hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, out, 1u, 0); }), 2);
    rep("all waves add-only (2/step)", time_it([&] { hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, out, 1u, 1); }), 2);
    rep("all waves mul-only (2/step)", time_it([&] { hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, out, 1u, 2); }), 2);
    return 0;
}
