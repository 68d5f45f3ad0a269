// valu_lab.hip — instruction-throughput microbenchmarks for the integer VALU ops the Poseidon2
// permutation is made of (MI355X / gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_lab valu_lab.hip
// Prints wave-instruction issue cycles per SIMD assuming every SIMD holds WAVES waves — twice (VERDICT r4 #5b):
//   wall-normalised   launch time (HIP events) x 2.4 GHz / wave-instructions per SIMD: what a step's wall time pays
//   shader cycles     clock64() (s_memtime: the shader clock) around each wave's loop / wave-instructions per SIMD
// and the clock the two imply (shader cycles of a wave / the launch's wall time).  "2.5 cycles" wall-normalised is
// 2.0 real cycles at 1.92 GHz or 2.5 real cycles at 2.4 GHz: the two columns tell which.  rocprofv3 --pmc GRBM_GUI_ACTIVE
// on this binary gives the same clock from the outside (tools/lab.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 65536
#define CHAINS 8

#define DEF_KERNEL(NAME, ASM_LINE)                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                      \
        uint32_t a[CHAINS];                                                                          \
        uint32_t b = seed | 1u, c = seed + 12345u;                                                   \
        uint64_t w[CHAINS];                                                                          \
        for (int i = 0; i < CHAINS; i++) { a[i] = threadIdx.x * 2654435761u + i + seed; w[i] = a[i]; } \
        const long long t0_ = clock64();                                                             \
        for (int it = 0; it < ITERS; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { ASM_LINE; }                         \
        }                                                                                            \
        const long long t1_ = clock64();                                                             \
        if ((threadIdx.x & 63u) == 0) ((long long*)out)[8 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1_ - t0_; \
        uint32_t r = 0;                                                                              \
        for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);         \
        if (r == 0x12345678u) out[0] = r;                                                            \
    }

DEF_KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_min, asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(b) : "s10", "s11"))
DEF_KERNEL(k_mad64_acc, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "s10", "s11"))
DEF_KERNEL(k_lshl_add64, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS])))
DEF_KERNEL(k_lshl_add64_0, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS])))
DEF_KERNEL(k_lshl64, asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(w[i])))
DEF_KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"))
DEF_KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20","s21"))
DEF_KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 15" : "+v"(a[i])))
DEF_KERNEL(k_pk_add16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_fma32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_fma64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS])))
DEF_KERNEL(k_mul_i32_i24, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))


DEF_KERNEL(k_sub, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_or, asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i])))
DEF_KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i])))
DEF_KERNEL(k_max, asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mini, asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)))
DEF_KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_add_lshl, asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_med3, asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_min3, asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEF_KERNEL(k_sub_co_cnd, asm volatile("v_sub_co_u32 %2, vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
DEF_KERNEL(k_addc, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEF_KERNEL(k_add_s, asm volatile("v_add_u32 %0, s20, %0" : "+v"(a[i]) : : "s20"))
DEF_KERNEL(k_add_lit, asm volatile("v_add_u32 %0, 0x80000001, %0" : "+v"(a[i])))
DEF_KERNEL(k_and_lit, asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(a[i])))
DEF_KERNEL(k_min_lit, asm volatile("v_min_u32 %0, 0x7fffffff, %0" : "+v"(a[i])))
DEF_KERNEL(k_mad64_s, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s20, %0" : "+v"(w[i]) : "v"(a[i]) : "s10", "s11", "s20"))
DEF_KERNEL(k_mul_lo_2dep, asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(a[i])))
DEF_KERNEL(k_sub_nc, asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_ashr, asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i])))
DEF_KERNEL(k_dot, asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))

// ---- round 3 lab list: candidate forms for fewer / cheaper instructions in the permutation
DEF_KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_dot4_u8, asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_pk_mul16, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_pk_mad16, asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL(k_add_sdwa, asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_add_dpp, asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_sub_clamp, asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_cvt_f64_u32, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(w[i]) : "v"(a[i])))
DEF_KERNEL(k_cvt_u32_f64, asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(w[i])))
DEF_KERNEL(k_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)))

// MFMA rows: one MFMA per "instruction" of the generic harness (8 independent accumulators per wave).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma_i8(uint32_t* out, uint32_t seed) {
    v4i acc[CHAINS];
    for (int i = 0; i < CHAINS; i++) acc[i] = v4i{(int)seed, 1, 2, (int)threadIdx.x};
    const int a = (int)(threadIdx.x * 2654435761u), b = (int)seed | 1;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a * 0x100000001L, (long)b * 0x100000001L, acc[i], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= acc[i][0] ^ acc[i][1] ^ acc[i][2] ^ acc[i][3];
    if (r == 0x12345678) out[0] = (uint32_t)r;
}
__global__ __launch_bounds__(256) void k_mfma_f64(uint32_t* out, uint32_t seed) {
    v4d acc[CHAINS];
    for (int i = 0; i < CHAINS; i++) acc[i] = v4d{(double)seed, 1.0, 2.0, (double)threadIdx.x};
    const double a = (double)threadIdx.x, b = (double)(seed | 1);
    for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double r = 0;
    for (int i = 0; i < CHAINS; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 0.12345678) out[0] = 1;
}
// Does the matrix pipe run BESIDE the VALU?  Odd workgroups issue only v_add_u32, even ones only MFMA i8: with the same
// total wave count as the pure rows, the time of this row against max(pure VALU, pure MFMA) / their sum tells.
__global__ __launch_bounds__(256) void k_mix_mfma_valu(uint32_t* out, uint32_t seed) {
    if (blockIdx.x & 1) {
        uint32_t a[CHAINS];
        uint32_t b = seed | 1u;
        for (int i = 0; i < CHAINS; i++) a[i] = threadIdx.x * 2654435761u + i + seed;
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
        uint32_t r = 0;
        for (int i = 0; i < CHAINS; i++) r ^= a[i];
        if (r == 0x12345678u) out[0] = r;
    } else {
        v4i acc[CHAINS];
        for (int i = 0; i < CHAINS; i++) acc[i] = v4i{(int)seed, 1, 2, (int)threadIdx.x};
        const int a = (int)(threadIdx.x * 2654435761u), b = (int)seed | 1;
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a * 0x100000001L, (long)b * 0x100000001L, acc[i], 0, 0, 0);
        }
        int r = 0;
        for (int i = 0; i < CHAINS; i++) r ^= acc[i][0] ^ acc[i][1] ^ acc[i][2] ^ acc[i][3];
        if (r == 0x12345678) out[0] = (uint32_t)r;
    }
}

// what hipcc emits behind an inline-asm statement whose result the next instruction reads (dst-sel forwarding hazard
// assumed for opaque asm on gfx940+): the same instruction followed by s_nop 0
DEF_KERNEL(k_add_nop, asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL(k_mad64_acc_nop, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\ts_nop 0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "s10", "s11"))
DEF_KERNEL(k_lshl_add64_nop, asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n\ts_nop 0" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS])))
// a dependent S-box-like chain: mad -> lshr -> add -> sub -> min, with and without the nops
DEF_KERNEL(k_chain, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, 0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, 0x80000001, %1\n\tv_min_u32 %1, %1, %2" : "+v"(w[i]), "+v"(a[i]), "+v"(c) : : "s10", "s11"))
DEF_KERNEL(k_chain_nop, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, 0\n\ts_nop 0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, 0x80000001, %1\n\tv_min_u32 %1, %1, %2\n\ts_nop 0" : "+v"(w[i]), "+v"(a[i]), "+v"(c) : : "s10", "s11"))

// ---- fully DEPENDENT chains (one chain per wave): what a wave costs the SIMD when every instruction reads the result
// of the one before it, with and without wait states in between.  Costs are per VALU instruction (nops not counted).
#define DEF_DEP(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                      \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = seed | 1u;                                \
        uint64_t w = a;                                                                              \
        const long long t0_ = clock64();                                                             \
        for (int it = 0; it < ITERS; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { BODY; }                             \
        }                                                                                            \
        const long long t1_ = clock64();                                                             \
        if ((threadIdx.x & 63u) == 0) ((long long*)out)[8 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1_ - t0_; \
        if ((a ^ (uint32_t)w ^ (uint32_t)(w >> 32)) == 0x12345678u) out[0] = a;                      \
    }
DEF_DEP(d_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)))
DEF_DEP(d_add_n0, asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(a) : "v"(b)))
DEF_DEP(d_add_n1, asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 1" : "+v"(a) : "v"(b)))
DEF_DEP(d_add_n3, asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 3" : "+v"(a) : "v"(b)))
DEF_DEP(d_min, asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b)))
DEF_DEP(d_min_n0, asm volatile("v_min_u32 %0, %0, %1\n\ts_nop 0" : "+v"(a) : "v"(b)))
DEF_DEP(d_min_n1, asm volatile("v_min_u32 %0, %0, %1\n\ts_nop 1" : "+v"(a) : "v"(b)))
DEF_DEP(d_mad, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0" : "+v"(w) : "v"(b) : "s10", "s11"))
DEF_DEP(d_mad_n0, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\ts_nop 0" : "+v"(w) : "v"(b) : "s10", "s11"))
DEF_DEP(d_mad_n1, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\ts_nop 1" : "+v"(w) : "v"(b) : "s10", "s11"))
DEF_DEP(d_mad_n3, asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\ts_nop 3" : "+v"(w) : "v"(b) : "s10", "s11"))
// add -> mad -> (64-bit add as the consumer of the product) -> back to add through a 64-bit operand
DEF_DEP(d_mix, asm volatile("v_add_u32 %1, %1, %2\n\tv_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\tv_lshl_add_u64 %0, %0, 1, %0" : "+v"(w), "+v"(a) : "v"(b) : "s10", "s11"))
DEF_DEP(d_mix_n0, asm volatile("v_add_u32 %1, %1, %2\n\ts_nop 0\n\tv_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\ts_nop 0\n\tv_lshl_add_u64 %0, %0, 1, %0\n\ts_nop 0" : "+v"(w), "+v"(a) : "v"(b) : "s10", "s11"))
DEF_DEP(d_mix_n1, asm volatile("v_add_u32 %1, %1, %2\n\ts_nop 1\n\tv_mad_u64_u32 %0, s[10:11], %1, %1, %0\n\ts_nop 1\n\tv_lshl_add_u64 %0, %0, 1, %0\n\ts_nop 1" : "+v"(w), "+v"(a) : "v"(b) : "s10", "s11"))

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; };

int main(int argc, char** argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    uint32_t* out;
    const size_t out_bytes = 64 + 8 * 4 * 256 * 8 * 2;  // [8 ..): shader cycles of every wave's loop (kernels of the two macros)
    hipMalloc(&out, out_bytes);
    std::vector<long long> clk((out_bytes - 64) / 8);
    Case cases[] = {{"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_min_u32", k_min}, {"v_and_b32", k_and},
                    {"v_alignbit_b32", k_alignbit}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add},
                    {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24},
                    {"v_mad_u32_u24", k_mad_u24}, {"v_mul_hi_u32_u24", k_mul_i32_i24},
                    {"v_mad_u64_u32(0)", k_mad64}, {"v_mad_u64_u32(acc)", k_mad64_acc},
                    {"v_lshl_add_u64<<1", k_lshl_add64}, {"v_lshl_add_u64<<0", k_lshl_add64_0}, {"v_lshlrev_b64", k_lshl64},
                    {"v_add_co_u32", k_add_co}, {"v_cndmask_b32", k_cndmask}, {"v_bfe_u32", k_bfe},
                    {"v_pk_add_u16", k_pk_add16}, {"v_fma_f32", k_fma32}, {"v_fma_f64", k_fma64},
                    {"v_sub_u32", k_sub}, {"v_subrev_u32", k_sub_nc}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_lshrrev_b32", k_lshr},
                    {"v_ashrrev_i32", k_ashr}, {"v_max_u32", k_max}, {"v_min_i32", k_mini}, {"v_mov_b32", k_mov}, {"v_bfi_b32", k_bfi}, {"v_xad_u32", k_xad},
                    {"v_add_lshl_u32", k_add_lshl}, {"v_med3_u32", k_med3}, {"v_min3_u32", k_min3}, {"cmp+cndmask (2)", k_cmp_cnd},
                    {"sub_co+cndmask (2)", k_sub_co_cnd}, {"v_addc_co_u32", k_addc}, {"v_add_u32 sgpr", k_add_s}, {"v_add_u32 literal", k_add_lit},
                    {"v_and_b32 literal", k_and_lit}, {"v_min_u32 literal", k_min_lit}, {"v_mad_u64_u32 sgpr", k_mad64_s}, {"v_mul_lo self", k_mul_lo_2dep}, {"v_mad_i32_i24", k_dot},
                    {"v_perm_b32", k_perm}, {"v_dot4_u32_u8", k_dot4_u8}, {"v_pk_mul_lo_u16", k_pk_mul16}, {"v_pk_mad_u16", k_pk_mad16},
                    {"v_add_u32_sdwa WORD_1", k_add_sdwa}, {"v_add_u32_dpp quad_perm", k_add_dpp}, {"v_sub_u32 clamp", k_sub_clamp},
                    {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_u32_f64", k_cvt_u32_f64}, {"v_min_f32", k_min_f32}, {"v_lshl_or_b32", k_lshl_or},
                    {"MFMA i32_16x16x32_i8", k_mfma_i8}, {"MFMA f64_16x16x4 (x1/4 iters)", k_mfma_f64},
                    {"MIX half waves v_add / half MFMA i8", k_mix_mfma_valu},
                    {"v_add_u32 + s_nop", k_add_nop}, {"v_mad_u64_u32(acc) + s_nop", k_mad64_acc_nop}, {"v_lshl_add_u64 + s_nop", k_lshl_add64_nop},
                    {"chain of 5", k_chain}, {"chain of 5 + 2 s_nop", k_chain_nop},
                    {"DEP v_add_u32", d_add}, {"DEP v_add_u32 + s_nop 0", d_add_n0}, {"DEP v_add_u32 + s_nop 1", d_add_n1}, {"DEP v_add_u32 + s_nop 3", d_add_n3},
                    {"DEP v_min_u32", d_min}, {"DEP v_min_u32 + s_nop 0", d_min_n0}, {"DEP v_min_u32 + s_nop 1", d_min_n1},
                    {"DEP v_mad_u64_u32", d_mad}, {"DEP v_mad_u64_u32 + s_nop 0", d_mad_n0}, {"DEP v_mad_u64_u32 + s_nop 1", d_mad_n1}, {"DEP v_mad_u64_u32 + s_nop 3", d_mad_n3},
                    {"DEP add,mad,add64 (3)", d_mix}, {"DEP add,mad,add64 + s_nop 0 each (3)", d_mix_n0}, {"DEP add,mad,add64 + s_nop 1 each (3)", d_mix_n1}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    printf("CUs %d, clock %d kHz, %d waves/SIMD\n", cus, prop.clockRate, waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-38s %9s %12s %12s %10s\n", "instruction", "ms", "cyc@2.4(wall)", "cyc(shader)", "clock GHz");
    for (auto& c : cases) {
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipMemset(out, 0, out_bytes);
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 1u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double insts_per_simd = 5.0 * waves_per_simd * (double)ITERS * CHAINS;  // wave-instructions per SIMD
        double ns_per_inst = ms * 1e6 / insts_per_simd;
        // shader cycles of the waves' loops (last launch): a wave shares its SIMD with waves_per_simd - 1 others for the whole loop
        hipMemcpy(clk.data(), (char*)out + 64, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
        // The SIMD's arbiter favours its oldest wave, so the waves of a SIMD finish one after the other (their MEAN lifetime
        // is ~5/8 of the launch at 4 waves per SIMD, ~9/16 at 8: measured, round 5) — the launch lasts as long as its
        // LONGEST-lived wave, and that one shares the SIMD for the whole loop: its cycles are the launch's shader cycles.
        double sum = 0, longest = 0; size_t cnt = 0;
        for (size_t k = 0; k < (size_t)blocks * 4; k++) if (clk[k] > 0) { sum += (double)clk[k]; cnt++; if ((double)clk[k] > longest) longest = (double)clk[k]; }
        if (cnt) {
            const double cyc_shader = longest / ((double)ITERS * CHAINS * waves_per_simd);   // per wave-instruction per SIMD
            const double ghz = longest / (ms / 5.0 * 1e6);                                   // shader cycles per ns of one launch
            printf("%-38s %9.3f %12.2f %12.2f %10.3f   (mean wave lifetime %.2f of the longest)\n", c.name, ms, ns_per_inst * 2.4, cyc_shader, ghz, sum / cnt / longest);
        } else printf("%-38s %9.3f %12.2f %12s %10s\n", c.name, ms, ns_per_inst * 2.4, "-", "-");
    }
    return 0;
}
