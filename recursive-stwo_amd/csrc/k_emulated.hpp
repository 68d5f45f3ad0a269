// k_emulated.hpp — SURVEY §8f.4: the gate values of the reference's EMULATED Poseidon2.
//
// poseidon_permute_emulated (primitives/poseidon31/src/emulated.rs:80-221) spells the permutation as gates of the
// Plonk-without-Poseidon constraint system — M4, pow5m4, pow5, Hadamard, grand-sum and add/mul rows
// (constraint_system/src/plonk_without_poseidon.rs:113-305) over QM31 variables that pack four state words — and
// every gate appends one variable to the circuit's witness (a pow5 gate two: the fourth powers are a witness of
// their own, emulated.rs:37-78).  This kernel produces those values: for each permutation, every variable the gadget
// appends once the circuit has its constants (they are cached by value, primitives/fields/src/qm31.rs:39-73, so only
// a circuit's first call allocates them), in allocation order.
//
//   rows [ 0, 12)   is_swap = Some((bit, _)) only (zero otherwise), emulated.rs:87-106, for i = 0, 1 each time:
//                   -left_i, right_i - left_i | (right_i - left_i) * bit | ... + left_i | -(...), right_i - (...)
//   rows [12, 23)   first linear layer :115: M4 of the four limbs (4), running sum (3), limb + sum (4)
//   then 4 full rounds x 19 rows :117-142   limb + constants (4) | per limb x^4, M4(x^5) (8) | sum (3) | limb + sum (4)
//   then 14 partial rounds x 17 rows :144-182   (x0,0,0,0), (0,x1,x2,x3), (x0 + rc,0,0,0), its ^4, its ^5, rejoined limb,
//                   grand sums of limbs 0+1 and 2+3, their sum (9) | per limb: limb o diagonal, sum + that (8)
//   then 4 full rounds x 19 rows :184-209                                               total 413 = RSV_EMU_ROWS
// Rows 409..412 are the permuted state; rows 413..415 are zero padding (RSV_EMU_STRIDE = 416 rows = 52 lines).
//
// MI355X mapping: HBM-write bound (65 B in, 6 656 B out per permutation; the arithmetic is ~1/8 of what the write
// costs).  One lane per permutation with the state in registers.  Rows are padded to RSV_EMU_STRIDE = 416 per
// permutation (three zero rows at the end) so that every permutation starts on a 128-byte line, and each wave stages
// its rows in an LDS ring ([64 lanes][16 rows + 1] x 16 B) that is drained EIGHT rows = one full aligned line per
// permutation at a time: a store instruction writes 8 complete lines (lanes 8k..8k+7 = the line of permutation k).
// Measured alternative: per-round flushes of 7-12 rows at 16-byte alignment reached 2.4 TB/s against 5.6 TB/s for
// this form (the lines at the two ends of every segment were written in two pieces a round apart, by which time
// the L2 had evicted the first).
#pragma once
#include "poseidon2.hpp"

namespace rsv {

constexpr int EMU_SWAP_ROWS = 12;
constexpr int EMU_ROWS = EMU_SWAP_ROWS + 401;      // 401 = is_swap None
constexpr int EMU_STRIDE = 416;                    // rows per permutation in memory: 52 lines of 128 bytes
constexpr int EMU_RING = 16;                       // staged rows per lane: drained in halves of 8
constexpr int EMU_PITCH = EMU_RING + 1;            // odd pitch: conflict-free b128 accesses

// Per-wave row sink.  head = index of the oldest staged row (always a multiple of 8), pending = rows staged.
// Callers put at most 8 rows between two drain() calls, so pending never exceeds 15.
struct EmuSink {
    uint4* my;        // this lane's ring
    const uint4* lds; // the wave's rings
    uint4* out;
    size_t p0, n;
    uint32_t head, pending;

    __device__ __forceinline__ void put(QM31 q) {
        my[(head + pending) & (EMU_RING - 1)] = make_uint4(q.a.a, q.a.b, q.b.a, q.b.b);
        pending++;
    }
    // One wave per workgroup: the barriers are the LDS hand-over between lanes of that wave.
    __device__ __forceinline__ void drain() {
        if (pending < 8) return;
        __syncthreads();
        const uint32_t half = head & 8;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t k = it * 64 + threadIdx.x, pl = k >> 3, row = k & 7;
            const uint4 v = lds[pl * EMU_PITCH + half + row];
            if (p0 + pl < n) out[(p0 + pl) * EMU_STRIDE + head + row] = v;
        }
        __syncthreads();
        head += 8;
        pending -= 8;
    }
};

__device__ __forceinline__ QM31 emu_m4(QM31 x) {
    mds4_ref(x.a.a, x.a.b, x.b.a, x.b.b);
    return x;
}
__device__ __forceinline__ uint32_t emu_pow4(uint32_t x) { return m_sqr(m_sqr(x)); }
__device__ __forceinline__ QM31 emu_select(bool c, QM31 x) { return c ? x : q_zero(); }

// One full round (emulated.rs:117-142 / :184-209): 4 + 8 + 7 rows.
__device__ __forceinline__ void emu_full_round(QM31* st, const uint32_t* rc, EmuSink& sink) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        st[i] = q_add(st[i], q_mk(rc[4 * i], rc[4 * i + 1], rc[4 * i + 2], rc[4 * i + 3]));
        sink.put(st[i]);
    }
    sink.drain();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const QM31 b = q_mk(emu_pow4(st[i].a.a), emu_pow4(st[i].a.b), emu_pow4(st[i].b.a), emu_pow4(st[i].b.b));
        sink.put(b);
        st[i] = emu_m4(q_mk(m_mul(st[i].a.a, b.a.a), m_mul(st[i].a.b, b.a.b), m_mul(st[i].b.a, b.b.a), m_mul(st[i].b.b, b.b.b)));
        sink.put(st[i]);
    }
    sink.drain();
    QM31 t = q_add(st[0], st[1]);
    sink.put(t);
    t = q_add(t, st[2]);
    sink.put(t);
    t = q_add(t, st[3]);
    sink.put(t);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        st[i] = q_add(st[i], t);
        sink.put(st[i]);
    }
    sink.drain();
}

// swap: NULL, or per permutation 0 = is_swap None, 1 = Some((false, _)), 2 = Some((true, _)); anything else, or an
// input word >= P, raises *bad (the rows of that permutation are then those of the all-zero input).
__global__ __launch_bounds__(64) void k_emulated(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                 const uint8_t* __restrict__ swap, uint4* __restrict__ out, size_t n,
                                                 uint32_t* __restrict__ bad) {
    __shared__ uint4 lds[64 * EMU_PITCH];
    const size_t p0 = (size_t)blockIdx.x * 64, p = p0 + threadIdx.x;
    const bool live = p < n;
    EmuSink sink{lds + threadIdx.x * EMU_PITCH, lds, out, p0, n, 0u, 0u};

    QM31 L[2] = {q_zero(), q_zero()}, R[2] = {q_zero(), q_zero()};
    uint32_t mode = 0;
    if (live) {
        const uint4* l4 = reinterpret_cast<const uint4*>(left + 8 * p);
        const uint4* r4 = reinterpret_cast<const uint4*>(right + 8 * p);
        uint32_t over = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint4 a = l4[i], b = r4[i];
            L[i] = q_mk(a.x, a.y, a.z, a.w);
            R[i] = q_mk(b.x, b.y, b.z, b.w);
            over |= (a.x >= P) | (a.y >= P) | (a.z >= P) | (a.w >= P) | (b.x >= P) | (b.y >= P) | (b.z >= P) | (b.w >= P);
        }
        if (swap) mode = swap[p];
        over |= mode > 2;
        if (over) {
            if (bad) atomicOr(bad, 1u);
            L[0] = L[1] = R[0] = R[1] = q_zero();  // keep the arithmetic below inside its stated ranges
            mode = 0;
        }
    }

    QM31 st[4];
    {   // emulated.rs:87-106: (right - left) * bit + left, right - (right - left) * bit
        const bool some = mode != 0, bit = mode == 2;
        QM31 d[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const QM31 nl = q_neg(L[i]);
            d[i] = q_add(R[i], nl);
            sink.put(emu_select(some, nl));
            sink.put(emu_select(some, d[i]));
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            d[i] = emu_select(bit, d[i]);
            sink.put(d[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            st[i] = q_add(d[i], L[i]);
            sink.put(emu_select(some, st[i]));
        }
        sink.drain();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const QM31 nd = q_neg(d[i]);
            st[2 + i] = q_add(R[i], nd);
            sink.put(emu_select(some, nd));
            sink.put(emu_select(some, st[2 + i]));
        }
    }
    {   // apply_16x16_mds_matrix, emulated.rs:24-35
        QM31 m[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            m[i] = emu_m4(st[i]);
            sink.put(m[i]);
        }
        sink.drain();
        QM31 t = q_add(m[0], m[1]);
        sink.put(t);
        t = q_add(t, m[2]);
        sink.put(t);
        t = q_add(t, m[3]);
        sink.put(t);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            st[i] = q_add(m[i], t);
            sink.put(st[i]);
        }
        sink.drain();
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) emu_full_round(st, RC_FULL[r], sink);
#pragma unroll 1
    for (int r = 0; r < 14; r++) {
        const QM31 x = st[0];
        sink.put(q_mk(x.a.a, 0, 0, 0));                       // Hadamard with the constant one
        sink.put(q_mk(0, x.a.b, x.b.a, x.b.b));               // Hadamard with (0, 1, 1, 1)
        const uint32_t y = m_add(x.a.a, RC_PARTIAL[r]);
        sink.put(q_mk(y, 0, 0, 0));
        const uint32_t y4 = emu_pow4(y), y5 = m_mul(y, y4);
        sink.put(q_mk(y4, 0, 0, 0));
        sink.put(q_mk(y5, 0, 0, 0));
        sink.drain();
        st[0] = q_mk(y5, x.a.b, x.b.a, x.b.b);
        sink.put(st[0]);
        const uint32_t s1 = m_add(m_add(m_add(st[0].a.a, st[0].a.b), m_add(st[0].b.a, st[0].b.b)),
                                  m_add(m_add(st[1].a.a, st[1].a.b), m_add(st[1].b.a, st[1].b.b)));
        const uint32_t s2 = m_add(m_add(m_add(st[2].a.a, st[2].a.b), m_add(st[2].b.a, st[2].b.b)),
                                  m_add(m_add(st[3].a.a, st[3].a.b), m_add(st[3].b.a, st[3].b.b)));
        const uint32_t s = m_add(s1, s2);
        sink.put(q_mk(s1, s1, s1, s1));
        sink.put(q_mk(s2, s2, s2, s2));
        sink.put(q_mk(s, s, s, s));
        sink.drain();
#pragma unroll
        for (int i = 0; i < 4; i++) {   // diagonal 3, 4, 8, ..., 65536 (parameters.rs:6-23): word k > 0 times 2^(k+1)
            QM31 h;
            h.a.a = i == 0 ? m_add(m_dbl(st[0].a.a), st[0].a.a) : m_shl(st[i].a.a, 4 * i + 1);
            h.a.b = m_shl(st[i].a.b, 4 * i + 2);
            h.b.a = m_shl(st[i].b.a, 4 * i + 3);
            h.b.b = m_shl(st[i].b.b, 4 * i + 4);
            sink.put(h);
            st[i] = q_add(q_mk(s, s, s, s), h);
            sink.put(st[i]);
        }
        sink.drain();
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) emu_full_round(st, RC_FULL[r], sink);
    // 413 rows so far; three zero rows complete the last line
    for (int i = EMU_ROWS; i < EMU_STRIDE; i++) sink.put(q_zero());
    sink.drain();
}

}  // namespace rsv
