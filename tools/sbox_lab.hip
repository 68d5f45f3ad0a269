// sbox_lab.hip — does the compiler's `s_nop 0` after every inline-asm statement cost anything?  The S-box layer of
// poseidon2.hpp (pow5 after canon(fold2)) against the same arithmetic written as ONE asm block per S-box.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../recursive-stwo_amd/csrc -o sbox_lab sbox_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "poseidon2.hpp"
using namespace rsv;

// x in C -> x^5 in L2, one asm block (no compiler-inserted nops inside)
__device__ __forceinline__ uint32_t pow5_block(uint32_t x) {
    uint32_t r;
    // fixed scratch registers (declared as clobbers): AMDGPU inline asm cannot name the halves of a 64-bit operand
    asm("v_add_u32 v62, %1, %1\n\t"
        "v_mad_u64_u32 v[60:61], s[10:11], v62, %1, 0\n\t"
        "v_lshrrev_b32 v60, 1, v60\n\t"
        "v_add_u32 v60, v60, v61\n\t"
        "v_add_u32 v61, 0x80000001, v60\n\t"
        "v_min_u32 v60, v60, v61\n\t"
        "v_add_u32 v61, v60, v60\n\t"
        "v_mad_u64_u32 v[60:61], s[10:11], v61, v60, 0\n\t"
        "v_lshrrev_b32 v60, 1, v60\n\t"
        "v_add_u32 v60, v60, v61\n\t"
        "v_add_u32 v61, 0x80000001, v60\n\t"
        "v_min_u32 v60, v60, v61\n\t"
        "v_mad_u64_u32 v[60:61], s[10:11], v62, v60, 0\n\t"
        "v_lshrrev_b32 v60, 1, v60\n\t"
        "v_add_u32 %0, v60, v61"
        : "=v"(r)
        : "v"(x)
        : "s10", "s11", "v60", "v61", "v62");
    return r;
}

// native C++ products (the compiler emits v_mad_u64_u32 itself), doubling through an EMPTY asm that only hides the
// second operand from the x + x -> x << 1 canonicalisation
__device__ __forceinline__ uint32_t dbl_native(uint32_t x) {
    uint32_t y = x;
    asm("" : "+v"(y));
    return x + y;
}
__device__ __forceinline__ uint32_t pow5_native(uint32_t x) {
    const uint32_t xx = dbl_native(x);
    uint32_t c2 = canon(fold2((uint64_t)xx * x));
    uint32_t c4 = canon(fold2((uint64_t)dbl_native(c2) * c2));
    return fold2((uint64_t)xx * c4);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_sbox(uint32_t* out, uint32_t seed, int iters) {
    uint32_t s[16];
    for (int i = 0; i < 16; i++) s[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) & 0x3fffffffu;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t x = canon(s[i]);
            s[i] = VARIANT == 0 ? pow5(x) : VARIANT == 1 ? pow5_block(x) : pow5_native(x);
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    const int iters = 2048;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int blocks = prop.multiProcessorCount * wps;
    uint32_t *o0, *o1;
    hipMalloc(&o0, blocks * 1024); hipMalloc(&o1, blocks * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 3; v++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_sbox<0>, dim3(blocks), dim3(256), 0, 0, o0, 7u, iters);
            else if (v == 1) hipLaunchKernelGGL(k_sbox<1>, dim3(blocks), dim3(256), 0, 0, o1, 7u, iters);
            else hipLaunchKernelGGL(k_sbox<2>, dim3(blocks), dim3(256), 0, 0, o1, 7u, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double sboxes_per_simd = (double)wps * iters * 16;
        printf("variant %d (%s): %.3f ms, %.1f cycles@2.4GHz per wave-S-box per SIMD\n", v, v == 0 ? "pow5()" : v == 1 ? "one asm block" : "native products", best,
               best * 1e-3 * 2.4e9 / sboxes_per_simd);
    }
    // same results?
    uint32_t* h0 = new uint32_t[blocks * 256]; uint32_t* h1 = new uint32_t[blocks * 256];
    hipMemcpy(h0, o0, blocks * 1024, hipMemcpyDeviceToHost); hipMemcpy(h1, o1, blocks * 1024, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < blocks * 256; i++) bad += h0[i] != h1[i];
    printf("mismatches: %zu\n", bad);
    return 0;
}
