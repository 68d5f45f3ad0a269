// perm_lab.hip — prototype / benchmark harness for Poseidon2-M31 permutation variants on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../recursive-stwo_amd/csrc -o perm_lab perm_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

#include "poseidon2.hpp"

using namespace rsv;

template <int VARIANT>
__global__ __launch_bounds__(256) void k_perm(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, int reps) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 v = in[4 * i + k];
        s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w;
    }
    for (int r = 0; r < reps; r++) {
        if (VARIANT == 0) poseidon2_ref_inline(s);
        else if (VARIANT == 1) poseidon2_inline(s);
        else {
            State16 st;
#pragma unroll
            for (int k = 0; k < 16; k++) st.s[k] = s[k];
            st = poseidon2(st);
#pragma unroll
            for (int k = 0; k < 16; k++) s[k] = st.s[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * i + k] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
}

int main(int argc, char** argv) {
    size_t n = 1 << 22;
    int reps = argc > 1 ? atoi(argv[1]) : 4;
    std::vector<uint32_t> h(16 * n);
    std::mt19937 rng(1);
    for (auto& w : h) w = rng() % 0x7fffffffu;
    for (int i = 0; i < 16; i++) h[i] = i;
    uint32_t *din, *d0, *d1;
    hipMalloc(&din, 64 * n); hipMalloc(&d0, 64 * n); hipMalloc(&d1, 64 * n);
    hipMemcpy(din, h.data(), 64 * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    for (int v = 0; v < 3; v++) {
        uint32_t* dout = v ? d1 : d0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_perm<0>, dim3(n / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)dout, n, reps);
            else if (v == 1) hipLaunchKernelGGL(k_perm<1>, dim3(n / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)dout, n, reps);
            else hipLaunchKernelGGL(k_perm<2>, dim3(n / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)dout, n, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
        printf("variant %d: %.3f ms for %zu x %d perms = %.3f G perms/s\n", v, ms[v], n, reps, n * (double)reps / ms[v] / 1e6);
    }
    {   // latency mode: exactly one wave per SIMD, 64 sequential permutations per lane
        size_t nl = 256 * 256;
        for (int v = 0; v < 3; v++) {
            float t = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_perm<0>, dim3(nl / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)d1, nl, 64);
                else if (v == 1) hipLaunchKernelGGL(k_perm<1>, dim3(nl / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)d1, nl, 64);
                else hipLaunchKernelGGL(k_perm<2>, dim3(nl / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)d1, nl, 64);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&t, e0, e1);
            }
            printf("latency mode variant %d: %.3f ms for 64 sequential perms/lane at 1 wave/SIMD = %.2f us per perm-step\n", v, t, t * 1000 / 64);
        }
        // restore d1 for the comparison below
        hipLaunchKernelGGL(k_perm<2>, dim3(n / 256), dim3(256), 0, 0, (const uint4*)din, (uint4*)d1, n, reps);
        hipDeviceSynchronize();
    }
    std::vector<uint32_t> o0(16 * n), o1(16 * n);
    hipMemcpy(o0.data(), d0, 64 * n, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), d1, 64 * n, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < 16 * n; i++) bad += o0[i] != o1[i];
    printf("mismatching words: %zu\n", bad);
    if (reps == 1) {
        const uint32_t kat[16] = {260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943,
                                  1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264};
        int ok0 = 1, ok1 = 1;
        for (int i = 0; i < 16; i++) { ok0 &= o0[i] == kat[i]; ok1 &= o1[i] == kat[i]; }
        printf("KAT: ref %s, new %s\n", ok0 ? "ok" : "FAIL", ok1 ? "ok" : "FAIL");
    }
    printf("speedup %.3fx\n", ms[0] / ms[1]);
    return bad != 0;
}
