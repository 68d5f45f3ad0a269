// circuit_program.hpp — HOST code, plain C++ (no HIP): everything rsv_witness_program_build / _create do that needs no
// device.  parse_template locates the sections of a template proof, build_program replays the mirrored gadgets
// (circuit_verifier.hpp) over it with the hints of a verifying pass and sorts what they did into a witness program,
// check_program validates a program before any kernel indexes with it.  circuit_builder.inc / witness_api.inc call these
// with the GPU's hints; tests/host_logic_asan.cpp calls them, built by g++ with AddressSanitizer + UBSan, with hints
// recorded on the CPU — the first sanitizer pass this host code has had.
#pragma once
#include <cstdio>
#include <cstring>

#include "../../include/rsv.h"
#include "layout.hpp"
#include "circuit_verifier.hpp"

namespace rsv::circuit {

constexpr uint32_t FIXED_PART_WORDS = make_sample_table().end;  // the constant-shape prefix of a proof (layout.hpp)

// The parts of a PlonkWithPoseidonProof the circuit allocates (bincode layout: SURVEY App. A), located on the host.
inline int parse_template(const uint8_t* proof, size_t len, Template& d) {
    // (the proof is read as 32-bit words: a buffer that is not word aligned is refused, not dereferenced)
    if (len < 4 * (FIXED_PART_WORDS + 8) || (len & 3) || (reinterpret_cast<uintptr_t>(proof) & 3)) return RSV_E_SIZE;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(proof);
    const size_t n_words = len / 4;
    d.w = w;
    d.lp = w[W_LP]; d.lq = w[W_LQ]; d.pow_bits = w[W_POW_BITS]; d.blowup = w[W_BLOWUP]; d.log_last = w[W_LOG_LAST]; d.nq = w[W_NQ];
    if (d.lp > 24 || d.lq > 24 || d.blowup > 16 || d.log_last > 16 || d.nq < 4 || d.nq > (uint32_t)MAXQ) return RSV_E_SIZE;
    d.A = d.lp + d.blowup; d.B = d.lq + d.blowup;
    d.M = std::max(d.lp + 1, d.lq + 2) + d.blowup;
    if (d.M > (uint32_t)MAX_LOG) return RSV_E_SIZE;
    constexpr SampleTable tbl = make_sample_table();
    d.sample_off.assign(tbl.off, tbl.off + N_SAMPLES);
    uint32_t k = 0;
    for (int t = 0; t < 4; t++) {
        d.samples_per_col[t].clear();
        for (uint32_t c = 0; c < tree_cols(t); c++) {
            std::vector<uint32_t> col;
            for (uint32_t s = 0; s < n_samples_of(t, (int)c); s++) col.push_back(k++);
            d.samples_per_col[t].push_back(col);
        }
    }
    size_t pos = tbl.end;
    auto need = [&](size_t words) { return pos + words <= n_words; };
    if (!need(2)) return RSV_E_SIZE;
    pos += 2;                                     // decommitments: 4
    for (int t = 0; t < 4; t++) {                 // hash_witness, column_witness
        if (!need(2)) return RSV_E_SIZE;
        const uint64_t nh = w[pos];
        pos += 2;
        if (!need(8 * nh + 2)) return RSV_E_SIZE;
        pos += 8 * nh + 2;
    }
    if (!need(2)) return RSV_E_SIZE;
    pos += 2;                                     // queried_values: 4
    for (int t = 0; t < 4; t++) {
        if (!need(2)) return RSV_E_SIZE;
        const uint64_t nv = w[pos];
        pos += 2;
        if (!need(nv)) return RSV_E_SIZE;
        pos += nv;
    }
    if (!need(2)) return RSV_E_SIZE;
    d.nonce_off = (uint32_t)pos;
    pos += 2;
    auto layer = [&](uint32_t& commit_off) {
        if (!need(2)) return false;
        const uint64_t nw = w[pos];
        pos += 2;
        if (!need(4 * nw + 2)) return false;
        pos += 4 * nw;
        const uint64_t nh = w[pos];
        pos += 2;
        if (!need(8 * nh + 2 + 8)) return false;
        pos += 8 * nh + 2;
        commit_off = (uint32_t)pos;
        pos += 8;
        return true;
    };
    if (!layer(d.first_commit_off)) return RSV_E_SIZE;
    if (!need(2)) return RSV_E_SIZE;
    d.n_inner = w[pos];
    pos += 2;
    if (d.n_inner > (uint32_t)MAX_INNER) return RSV_E_SIZE;
    for (uint32_t l = 0; l < d.n_inner; l++)
        if (!layer(d.inner_commit_off[l])) return RSV_E_SIZE;
    if (!need(2)) return RSV_E_SIZE;
    d.last_n = w[pos];
    pos += 2;
    d.last_off = (uint32_t)pos;
    if (d.last_n != (1u << d.log_last) || !need(4ull * d.last_n + 1)) return RSV_E_SIZE;
    return RSV_OK;
}


// Everything a level kernel will index with comes from the program, so the program is checked once, here: a bad one is
// an API error, never a device fault.
inline int check_program(const uint32_t* instr, size_t n_instr, const uint32_t* level_offsets, size_t n_levels, uint32_t n_vars,
                         const rsv_witness_shape& s, bool log = false) {
    if (n_instr != n_vars || n_levels == 0 || level_offsets[0] != 0 || level_offsets[n_levels] != n_instr) return RSV_E_SIZE;
    if (s.n_queries < 4 || s.n_queries > (uint32_t)MAXQ || s.n_inner > (uint32_t)MAX_INNER || s.log_last > 16 || s.flow_count == 0)
        return RSV_E_SIZE;
    const uint32_t M = (s.log_size_plonk + 1 > s.log_size_poseidon + 2 ? s.log_size_plonk + 1 : s.log_size_poseidon + 2) + s.log_blowup;
    if (M > (uint32_t)MAX_LOG) return RSV_E_SIZE;
    std::vector<uint32_t> level_of(n_vars, 0xFFFFFFFFu);
    for (size_t l = 0; l < n_levels; l++) {
        if (level_offsets[l] > level_offsets[l + 1]) return RSV_E_SIZE;
        for (uint32_t k = level_offsets[l]; k < level_offsets[l + 1]; k++) {
            const uint32_t* in = instr + (size_t)k * 8;
            const uint32_t op = in[0], dst = in[1], a = in[2], b = in[3], i0 = in[4], i1 = in[5], i2 = in[6];
            if (op >= W_N_OPS || dst >= n_vars || level_of[dst] != 0xFFFFFFFFu) {
                if (log) fprintf(stderr, "[rsv] witness program: instruction %u has op %u, destination %u (of %u variables)\n", k, op, dst, n_vars);
                return RSV_E_RANGE;
            }
            auto before = [&](uint32_t v) { return v < n_vars && level_of[v] < l; };  // produced by an earlier level
            bool ok = true;
            switch (op) {
            case W_CONST: ok = i0 < MP && i1 < MP && i2 < MP && in[7] < MP; break;
            case W_ADD: case W_MUL: ok = before(a) && before(b); break;
            case W_MULC: ok = before(a) && i0 < MP; break;
            case W_COPY: case W_INV: case W_INV0: case W_QINV: ok = before(a); break;
            case W_CINV: ok = before(a) && i0 < 2; break;
            case W_COORD: ok = before(a) && i0 < 4; break;
            case W_BIT: ok = before(a) && i0 < 31; break;
            case W_FLOW: ok = i0 < s.flow_count && i1 >= 16 && i1 < 32 && (i1 & 3) == 0; break;
            case W_WORD: ok = i0 < FIXED_PART_WORDS; break;
            case W_WORD4: ok = i0 + 4 <= FIXED_PART_WORDS; break;
            case W_FRI_COMMIT: ok = i0 <= s.n_inner && i1 < 2; break;
            case W_LAST_POLY: ok = i0 < (1u << s.log_last); break;
            case W_NONCE: ok = i0 < 3; break;
            case W_TRACE_COL: ok = i0 < 4 && i1 < s.n_queries && i2 < 64; break;
            case W_FRI_COL: ok = i0 <= s.n_inner && i1 < s.n_queries && i2 < 24 && (i2 & 3) == 0; break;
            default: ok = false;
            }
            if (!ok) { if (log) fprintf(stderr, "[rsv] witness program: instruction %u (op %u, dst %u, a %u, b %u, imm %u %u %u) is out of range\n", k, op, dst, a, b, i0, i1, i2); return RSV_E_RANGE; }
        }
        for (uint32_t k = level_offsets[l]; k < level_offsets[l + 1]; k++) level_of[instr[(size_t)k * 8 + 1]] = (uint32_t)l;
    }
    return RSV_OK;
}


// What a run of the gadgets leaves behind: the program (one instruction per variable, sorted by dependency depth), the
// wire indices of every Poseidon invocation, the gate list and the rows whose `op` follows the witness.
struct BuiltProgram {
    size_t n_vars = 0;
    std::vector<uint32_t> instr, level_offsets, flow_wires, gates, witness_ops;
};

// d: the parsed template with its hint pointers set; flow / swap: the PoseidonFlow of ONE copy of the verifier.
inline int build_program(const Template& d, const uint32_t* flow, const uint8_t* swap, uint32_t flow_count,
                         const std::vector<std::pair<uint32_t, Q4>>& inputs, uint32_t copies, const uint8_t* set_walks, BuiltProgram& out,
                         bool log = false) {
    ConstraintSystem cs;
    FlowSource fsrc{flow, swap, flow_count, 0};
    Gadgets g{&cs, &fsrc};
    try {
        for (uint32_t c = 0; c < copies; c++) verify_in_circuit(g, d, inputs, set_walks ? (set_walks[c] & 3u) : 0u);
    } catch (const std::exception& e) {
        if (log) fprintf(stderr, "[rsv] witness program build: %s\n", e.what());
        return RSV_E_RANGE;
    }
    if (fsrc.cursor != (size_t)copies * flow_count) return RSV_E_RANGE;

    // one instruction per variable, sorted by dependency depth
    const size_t n_vars = cs.variables.size();
    std::vector<uint32_t> depth(n_vars, 0);
    uint32_t max_depth = 0;
    for (size_t k = 0; k < n_vars; k++) {
        const Instr& in = cs.origin[k];
        uint32_t dep = 0;
        switch (in.op) {
        case W_ADD: case W_MUL: dep = 1 + std::max(depth[in.a], depth[in.b]); break;
        case W_MULC: case W_COPY: case W_INV: case W_INV0: case W_QINV: case W_CINV: case W_COORD: case W_BIT: dep = 1 + depth[in.a]; break;
        default: break;
        }
        depth[k] = dep;
        max_depth = std::max(max_depth, dep);
    }
    std::vector<uint32_t> level_offsets(max_depth + 2, 0);
    for (size_t k = 0; k < n_vars; k++) level_offsets[depth[k] + 1]++;
    for (size_t l = 0; l + 1 < level_offsets.size(); l++) level_offsets[l + 1] += level_offsets[l];
    std::vector<uint32_t> cursor(level_offsets.begin(), level_offsets.end() - 1), instr(n_vars * 8);
    for (size_t k = 0; k < n_vars; k++) {  // stable: ascending variable index inside a level
        const Instr& in = cs.origin[k];
        uint32_t* o = &instr[(size_t)cursor[depth[k]]++ * 8];
        o[0] = in.op; o[1] = in.dst; o[2] = in.a; o[3] = in.b; o[4] = in.imm[0]; o[5] = in.imm[1]; o[6] = in.imm[2]; o[7] = in.imm[3];
    }
    out.n_vars = n_vars;
    out.instr = std::move(instr);
    out.level_offsets = std::move(level_offsets);
    out.flow_wires.reserve(cs.flow.size() * 5);
    for (const FlowRecord& f : cs.flow) {
        for (int j = 0; j < 4; j++) out.flow_wires.push_back(f.wire[j]);
        out.flow_wires.push_back(f.addr);
    }
    out.gates.reserve(cs.rows.size() * 6);
    for (const GateRow& r : cs.rows)
        for (uint32_t v : {r.a, r.b, r.c, r.op, r.poseidon_wire, r.enforce_c_m31}) out.gates.push_back(v);
    for (const WitnessOp3& wo : cs.witness_ops)
        for (uint32_t v : {wo.row, wo.bit, wo.constant}) out.witness_ops.push_back(v);
    return RSV_OK;
}

}  // namespace rsv::circuit
