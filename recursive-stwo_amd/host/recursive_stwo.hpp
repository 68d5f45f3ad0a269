// recursive_stwo.hpp — C++ host-side mirror of the reference's interface for the verify path.
//
// Rust is not available in this image, so the host layer above the C-ABI (include/rsv.h) is C++.
// Names, argument meaning and failure behaviour follow the reference's Rust items (values only —
// the gate-recording half of every *Var type is out of scope):
//
//   poseidon31::poseidon2_permute          primitives/poseidon31/src/implementation.rs:108-149
//   Poseidon2HalfVar::{permute, permute_get_rate, permute_get_capacity, swap_permute_get_*}
//                                          primitives/poseidon31/src/lib.rs:251-423
//   poseidon31::emulated::poseidon_permute_emulated{,_batch}   (witness values of the gate-level permutation)
//                                          primitives/poseidon31/src/emulated.rs:80-221
//   Poseidon31MerkleHasherVar::*           primitives/merkle/src/lib.rs:9-181
//   ChannelVar::{mix_root, draw_felts, mix_one_felt, mix_two_felts}
//                                          primitives/channel/src/lib.rs:24-58
//   SinglePathMerkleProof::verify          components/hints/src/decommit.rs:22-42
//   FiatShamirResults::compute             components/recursive/fiat_shamir/src/lib.rs:31-176
//   verify (FiatShamir → Composition → Answer → Folding)   examples/single-proof/src/main.rs:48-82
//   hints::{DecommitHints, SinglePairMerkleProof, FirstLayerHints, InnerLayersHints}::compute
//                                          components/hints/src/decommit.rs:186-250, folding.rs:21-91,290-601
//
// The reference panics (assert_eq!/unwrap, panic = 'abort') when a check fails; here a failed check
// throws recursive_stwo::VerificationError carrying the stage, and API/device failures throw
// recursive_stwo::DeviceError.  Every call runs on the GPU through librsv_hip.so; there is no CPU path.
// These single-item wrappers exist for drop-in parity; throughput comes from Verifier::verify_batch.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rsv.h"

namespace recursive_stwo {

struct DeviceError : std::runtime_error {
    int status;
    DeviceError(const std::string& what, int st) : std::runtime_error(what + ": rsv_status " + std::to_string(st)), status(st) {}
};
struct VerificationError : std::runtime_error {
    rsv_reason reason;
    explicit VerificationError(rsv_reason r) : std::runtime_error("proof rejected at stage " + std::to_string((int)r)), reason(r) {}
};
inline void check(int st, const char* what) {
    if (st != RSV_OK) throw DeviceError(what, st);
}

using M31 = uint32_t;                      // canonical word < 2^31 - 1
using QM31 = std::array<uint32_t, 4>;      // (a0, a1, b0, b1)
using Hash = std::array<uint32_t, 8>;      // Poseidon31Hash

inline int& default_device() {
    static int d = 0;
    return d;
}

namespace poseidon31 {
// pub fn poseidon2_permute(p_state: &mut [M31; 16])
inline void poseidon2_permute(std::array<M31, 16>& state) {
    std::array<M31, 16> out{};
    check(rsv_poseidon2_permute(state.data(), out.data(), 1, default_device()), "rsv_poseidon2_permute");
    state = out;
}
}  // namespace poseidon31

// Poseidon2HalfVar: one 8-word half of the sponge state.
struct Poseidon2HalfVar {
    Hash value{};
    static Poseidon2HalfVar zero() { return {}; }
    static Poseidon2HalfVar from_m31(const M31* v8) {
        Poseidon2HalfVar h;
        for (int i = 0; i < 8; i++) h.value[i] = v8[i];
        return h;
    }
    static Poseidon2HalfVar from_qm31(const QM31& a, const QM31& b) {
        Poseidon2HalfVar h;
        for (int i = 0; i < 4; i++) { h.value[i] = a[i]; h.value[4 + i] = b[i]; }
        return h;
    }
    std::array<QM31, 2> to_qm31() const {
        return {QM31{value[0], value[1], value[2], value[3]}, QM31{value[4], value[5], value[6], value[7]}};
    }
    // pub fn permute(left, right, ignore_left_result, ignore_right_result, is_swap) -> (Half, Half)
    static std::pair<Poseidon2HalfVar, Poseidon2HalfVar> permute(const Poseidon2HalfVar& left, const Poseidon2HalfVar& right,
                                                                 bool /*ignore_left_result*/, bool /*ignore_right_result*/,
                                                                 std::optional<bool> is_swap) {
        Poseidon2HalfVar rate, cap;
        uint8_t sw = is_swap.value_or(false) ? 1 : 0;
        check(rsv_poseidon2_half_permute(left.value.data(), right.value.data(), is_swap ? &sw : nullptr, rate.value.data(),
                                         cap.value.data(), 1, default_device()),
              "rsv_poseidon2_half_permute");
        return {rate, cap};
    }
    static Poseidon2HalfVar permute_get_rate(const Poseidon2HalfVar& l, const Poseidon2HalfVar& r) {
        return permute(l, r, false, true, std::nullopt).first;
    }
    static Poseidon2HalfVar permute_get_capacity(const Poseidon2HalfVar& l, const Poseidon2HalfVar& r) {
        return permute(l, r, true, false, std::nullopt).second;
    }
    static Poseidon2HalfVar swap_permute_get_rate(const Poseidon2HalfVar& l, const Poseidon2HalfVar& r, bool is_swap) {
        return permute(l, r, false, true, is_swap).first;
    }
    static Poseidon2HalfVar swap_permute_get_capacity(const Poseidon2HalfVar& l, const Poseidon2HalfVar& r, bool is_swap) {
        return permute(l, r, true, false, is_swap).second;
    }
    // equalverify: the reference asserts; here a mismatch throws.
    void equalverify(const Poseidon2HalfVar& rhs) const {
        if (value != rhs.value) throw VerificationError(RSV_R_MERKLE_T0);
    }
};
using HashVar = Poseidon2HalfVar;

namespace poseidon31::emulated {
// What one call of poseidon_permute_emulated appends to a Plonk-without-Poseidon circuit's `variables` once its
// constants are cached (primitives/poseidon31/src/emulated.rs:80-221): `swap_rows` (12, only for is_swap = Some),
// then `rows` (401); `out_left` / `out_right` are the values of the returned Poseidon2HalfEmulatedVar pair.
struct EmulatedWitness {
    std::vector<QM31> swap_rows, rows;
    std::array<QM31, 2> out_left{}, out_right{};
};
using IsSwap = std::optional<bool>;  // the bit's value; its variable index stays with the caller's circuit

// Batch form: one GPU launch for all permutations (rsv_poseidon2_emulated).
inline std::vector<EmulatedWitness> poseidon_permute_emulated_batch(const std::vector<std::array<QM31, 2>>& left,
                                                                     const std::vector<std::array<QM31, 2>>& right,
                                                                     const std::vector<IsSwap>& is_swap) {
    const size_t n = left.size();
    if (right.size() != n || is_swap.size() != n) throw DeviceError("poseidon_permute_emulated_batch: ragged input", RSV_E_SIZE);
    std::vector<uint32_t> l(8 * n), r(8 * n), rows((size_t)RSV_EMU_STRIDE * 4 * n);
    std::vector<uint8_t> sw(n);
    for (size_t p = 0; p < n; p++) {
        for (int i = 0; i < 8; i++) { l[8 * p + i] = left[p][i / 4][i % 4]; r[8 * p + i] = right[p][i / 4][i % 4]; }
        sw[p] = !is_swap[p] ? 0 : (*is_swap[p] ? 2 : 1);
    }
    check(rsv_poseidon2_emulated(l.data(), r.data(), sw.data(), rows.data(), n, default_device()), "rsv_poseidon2_emulated");
    std::vector<EmulatedWitness> out(n);
    for (size_t p = 0; p < n; p++) {
        auto row = [&](int k) { const uint32_t* q = &rows[((size_t)RSV_EMU_STRIDE * p + k) * 4]; return QM31{q[0], q[1], q[2], q[3]}; };
        if (is_swap[p]) for (int k = 0; k < RSV_EMU_SWAP_ROWS; k++) out[p].swap_rows.push_back(row(k));
        for (int k = RSV_EMU_SWAP_ROWS; k < RSV_EMU_ROWS; k++) out[p].rows.push_back(row(k));
        out[p].out_left = {row(RSV_EMU_ROWS - 4), row(RSV_EMU_ROWS - 3)};
        out[p].out_right = {row(RSV_EMU_ROWS - 2), row(RSV_EMU_ROWS - 1)};
    }
    return out;
}
// pub fn poseidon_permute_emulated(left, right, is_swap) -> (Poseidon2HalfEmulatedVar, Poseidon2HalfEmulatedVar)
inline EmulatedWitness poseidon_permute_emulated(const std::array<QM31, 2>& left, const std::array<QM31, 2>& right, IsSwap is_swap) {
    return poseidon_permute_emulated_batch({left}, {right}, {is_swap})[0];
}
}  // namespace poseidon31::emulated

// Poseidon31MerkleHasherVar (primitives/merkle/src/lib.rs)
struct Poseidon31MerkleHasherVar {
    static HashVar hash_node(const HashVar* left, const HashVar* right, const std::vector<M31>& cols) {
        HashVar out;
        check(rsv_merkle_hash_node(left ? left->value.data() : nullptr, right ? right->value.data() : nullptr,
                                   cols.empty() ? nullptr : cols.data(), cols.size(), out.value.data(), 1, default_device()),
              "rsv_merkle_hash_node");
        return out;
    }
    static HashVar hash_tree(const HashVar& l, const HashVar& r) { return hash_node(&l, &r, {}); }
    static HashVar hash_tree_with_swap(const HashVar& l, const HashVar& r, bool bit) { return bit ? hash_tree(r, l) : hash_tree(l, r); }
    static HashVar combine_hash_tree_with_column(const HashVar& tree, const HashVar& col) { return HashVar::permute_get_rate(tree, col); }
    static HashVar hash_tree_with_column(const HashVar& l, const HashVar& r, const HashVar& col) {
        return combine_hash_tree_with_column(hash_tree(l, r), col);
    }
    static HashVar hash_tree_with_column_hash_with_swap(const HashVar& l, const HashVar& r, bool bit, const HashVar& col) {
        return combine_hash_tree_with_column(hash_tree_with_swap(l, r, bit), col);
    }
    static HashVar hash_m31_columns_get_rate(const std::vector<M31>& m31) { return hash_node(nullptr, nullptr, m31); }
    static HashVar hash_m31_columns_get_capacity(const std::vector<M31>& m31) {
        HashVar d = HashVar::zero();
        for (size_t off = 0; off < m31.size(); off += 8) {
            M31 chunk[8] = {0};
            for (size_t i = 0; i < 8 && off + i < m31.size(); i++) chunk[i] = m31[off + i];
            d = HashVar::permute_get_capacity(HashVar::from_m31(chunk), d);
        }
        return d;
    }
    static HashVar hash_qm31_columns_get_capacity(const std::vector<QM31>& q) {
        std::vector<M31> flat;
        for (auto& v : q) flat.insert(flat.end(), v.begin(), v.end());
        return hash_m31_columns_get_capacity(flat);
    }
    static HashVar hash_qm31_columns_get_rate(const std::vector<QM31>& q) {
        return HashVar::permute_get_rate(HashVar::zero(), hash_qm31_columns_get_capacity(q));
    }
};

// ChannelVar (primitives/channel/src/lib.rs)
struct ChannelVar {
    size_t n_sent = 0;
    HashVar digest = HashVar::zero();
    void mix_root(const HashVar& root) { digest = HashVar::permute_get_capacity(root, digest); n_sent = 0; }
    std::array<QM31, 2> draw_felts() {
        QM31 n{(uint32_t)n_sent, 0, 0, 0};
        n_sent += 1;
        return HashVar::permute_get_rate(HashVar::from_qm31(n, QM31{0, 0, 0, 0}), digest).to_qm31();
    }
    void mix_one_felt(const QM31& f) { mix_two_felts(f, QM31{0, 0, 0, 0}); }
    void mix_two_felts(const QM31& f, const QM31& g) {
        digest = HashVar::permute_get_capacity(HashVar::from_qm31(f, g), digest);
        n_sent = 0;
    }
};

// SinglePathMerkleProof (components/hints/src/decommit.rs:10-42)
struct SinglePathMerkleProof {
    uint32_t query = 0;
    std::vector<Hash> sibling_hashes;                 // leaf level first
    std::vector<std::vector<M31>> columns;            // columns[h] for h = 0..depth (empty where none)
    Hash root{};
    uint32_t depth = 0;
    // pub fn verify(&self): panics on mismatch in the reference, throws here.
    void verify() const {
        std::vector<uint32_t> n_cols_at(depth + 1), cols;
        for (uint32_t h = 0; h <= depth; h++) n_cols_at[h] = (uint32_t)columns[h].size();
        for (uint32_t lvl = 0; lvl <= depth; lvl++) cols.insert(cols.end(), columns[depth - lvl].begin(), columns[depth - lvl].end());
        std::vector<uint32_t> sib;
        for (auto& h : sibling_hashes) sib.insert(sib.end(), h.begin(), h.end());
        Hash out{};
        check(rsv_merkle_path_root(&query, sib.data(), cols.data(), n_cols_at.data(), depth, out.data(), 1, default_device()),
              "rsv_merkle_path_root");
        if (out != root) throw VerificationError(RSV_R_MERKLE_T0);
    }
};

struct FriConfig {
    uint32_t log_last_layer_degree_bound, log_blowup_factor, n_queries;
    static FriConfig make(uint32_t log_last, uint32_t log_blowup, uint32_t n_queries) { return {log_last, log_blowup, n_queries}; }
};
struct PcsConfig {
    uint32_t pow_bits;
    FriConfig fri_config;
    rsv_pcs_config abi() const {
        return {pow_bits, fri_config.log_blowup_factor, fri_config.log_last_layer_degree_bound, fri_config.n_queries};
    }
};
using Inputs = std::vector<std::pair<uint32_t, QM31>>;  // &[(usize, QM31Var)]

inline std::vector<rsv_public_input> abi_inputs(const Inputs& inputs) {
    std::vector<rsv_public_input> pi(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) {
        pi[i].idx = inputs[i].first;
        for (int k = 0; k < 4; k++) pi[i].value[k] = inputs[i].second[k];
    }
    return pi;
}

// FiatShamirResults (components/recursive/fiat_shamir/src/lib.rs:13-28): the challenges of one proof.
struct FiatShamirResults {
    QM31 z, alpha, random_coeff, oods_t, oods_x, oods_y, after_sampled_values_random_coeff;
    std::vector<QM31> fri_alphas;
    std::vector<M31> raw_queries;
    uint32_t max_first_layer_column_log_size = 0;
    // pub fn compute(hints, proof, pcs_config, inputs): PoW failure panics in the reference, throws here.
    static FiatShamirResults compute(const std::vector<uint8_t>& proof) {
        std::vector<uint32_t> out(1024);
        check(rsv_transcript(proof.data(), proof.size(), out.data(), out.size(), default_device()), "rsv_transcript");
        if (out[0] != RSV_R_OK) throw VerificationError((rsv_reason)out[0]);
        FiatShamirResults r;
        auto q = [&](size_t o) { return QM31{out[o], out[o + 1], out[o + 2], out[o + 3]}; };
        uint32_t na = out[1], nq = out[2];
        r.max_first_layer_column_log_size = out[3];
        r.z = q(4); r.alpha = q(8); r.random_coeff = q(12); r.oods_t = q(16); r.oods_x = q(20); r.oods_y = q(24);
        r.after_sampled_values_random_coeff = q(28);
        for (uint32_t i = 0; i < na; i++) r.fri_alphas.push_back(q(40 + 4 * i));
        r.raw_queries.assign(out.begin() + 40 + 4 * na, out.begin() + 40 + 4 * na + nq);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------------
// components/hints: what the reference derives on the host with stwo's native verifier, here read from ONE
// verifying pass on the GPU (rsv_verify_hints).  A proof that does not verify throws VerificationError, as the
// reference's hint constructors panic.
// ---------------------------------------------------------------------------------------------------
// SinglePairMerkleProof (components/hints/src/folding.rs:21-91)
struct SinglePairMerkleProof {
    uint32_t query = 0;
    std::vector<Hash> sibling_hashes;
    std::vector<std::optional<QM31>> self_columns, siblings_columns;  // indexed by level h = 0..depth
    Hash root{};
    uint32_t depth = 0;
    static std::vector<M31> words(const std::optional<QM31>& q) { return q ? std::vector<M31>(q->begin(), q->end()) : std::vector<M31>{}; }
    // pub fn verify(&self)
    void verify() const {
        using H = Poseidon31MerkleHasherVar;
        HashVar self_hash = H::hash_node(nullptr, nullptr, words(self_columns[depth]));
        HashVar sibling_hash = H::hash_node(nullptr, nullptr, words(siblings_columns[depth]));
        for (uint32_t i = 0; i < depth; i++) {
            const uint32_t h = depth - i - 1;
            const bool bit = (query >> i) & 1u;
            const HashVar& l = bit ? sibling_hash : self_hash;
            const HashVar& r = bit ? self_hash : sibling_hash;
            if (!self_columns[h]) {
                self_hash = H::hash_node(&l, &r, {});
                if (i != depth - 1) sibling_hash = HashVar::from_m31(sibling_hashes[i].data());
            } else {
                self_hash = H::hash_node(&l, &r, words(self_columns[h]));
                HashVar column_hash = H::hash_m31_columns_get_capacity(words(siblings_columns[h]));
                sibling_hash = HashVar::permute_get_rate(HashVar::from_m31(sibling_hashes[i].data()), column_hash);
            }
        }
        Hash got{};
        for (int k = 0; k < 8; k++) got[k] = self_hash.value[k];
        if (got != root) throw VerificationError(RSV_R_FRI_FIRST);
    }
};

struct ProofShape {  // read from the proof header
    uint32_t lp, lq, log_blowup, log_last, n_queries, A, B, M, n_inner;
    static ProofShape of(const std::vector<uint8_t>& proof) {
        if (proof.size() < 64) throw VerificationError(RSV_R_PARSE);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(proof.data());
        ProofShape s{};
        s.lp = w[0]; s.lq = w[1]; s.log_blowup = w[11]; s.log_last = w[12]; s.n_queries = w[13];
        if (s.lp > 24 || s.lq > 24 || s.log_blowup > 8 || s.n_queries < 4 || s.n_queries > 128) throw VerificationError(RSV_R_PARSE);
        s.A = s.lp + s.log_blowup; s.B = s.lq + s.log_blowup;
        s.M = (s.lp + 1 > s.lq + 2 ? s.lp + 1 : s.lq + 2) + s.log_blowup;
        if (s.M < s.log_blowup + s.log_last + 1) throw VerificationError(RSV_R_PARSE);
        s.n_inner = s.M - s.log_blowup - s.log_last - 1;
        return s;
    }
    static uint32_t plonk_cols(int t) { return t == 0 ? 10u : t == 1 ? 12u : t == 2 ? 8u : 0u; }
    static uint32_t poseidon_cols(int t) { return t == 0 ? 40u : t == 1 ? 48u : t == 2 ? 8u : 0u; }
};

// PoseidonFlow (stwo_examples::plonk_with_poseidon::poseidon, used by constraint_system/src/plonk_with_poseidon.rs:13-15,
// 36,117-128): what the recursion circuit records for its Poseidon accelerator — one (entry_1, entry_2, entry_3,
// entry_4, swap_option) per Poseidon2HalfVar::permute invocation, in invocation order.  compute() takes the VALUE side
// from the GPU's verifying pass (rsv_hints_out::d_flow); the wire indices (PoseidonEntry::wire, SwapOption::addr) are
// circuit bookkeeping of the Rust constraint system and are 0 here.
struct PoseidonEntry {
    size_t wire = 0;
    Hash hash{};
};
struct SwapOption {
    size_t addr = 0;
    bool swap = false;
};
struct PoseidonFlow {
    struct Invocation { PoseidonEntry entry_1, entry_2, entry_3, entry_4; SwapOption swap_option; };
    std::vector<Invocation> invocations;  // = PoseidonFlow.0

    // the circuit of examples/multi-proofs/src/main.rs:69-139 for ONE verification of `proof`
    static PoseidonFlow compute(const std::vector<uint8_t>& proof, const PcsConfig& config, const Inputs& inputs) {
        const ProofShape sh = ProofShape::of(proof);
        const rsv_pcs_config abi_cfg = config.abi();
        uint32_t count = 0;
        if (rsv_poseidon_flow_count(sh.lp, sh.lq, &abi_cfg, &count) != RSV_OK) throw VerificationError(RSV_R_PARSE);
        std::vector<uint32_t> rec((size_t)count * 32), got(1);
        std::vector<uint8_t> swap(count);
        rsv_hints_out ho{};
        ho.d_flow = rec.data(); ho.d_flow_swap = swap.data(); ho.d_flow_count = got.data(); ho.flow_stride = count;
        const uint64_t offsets[2] = {0, proof.size()};
        uint8_t accept = 0, reason = 0;
        auto pi = abi_inputs(inputs);
        const rsv_cfg_set cfg_set{&abi_cfg, 1, nullptr};
        check(rsv_verify_hints(proof.data(), offsets, 1, &cfg_set, pi.data(), pi.size(), &ho, &accept, &reason, default_device()),
              "rsv_verify_hints");
        if (!accept) throw VerificationError((rsv_reason)reason);
        if (got[0] != count) throw VerificationError(RSV_R_PARSE);
        PoseidonFlow f;
        f.invocations.resize(count);
        for (uint32_t i = 0; i < count; i++) {
            Invocation& v = f.invocations[i];
            for (int k = 0; k < 8; k++) {
                v.entry_1.hash[k] = rec[(size_t)i * 32 + k];
                v.entry_2.hash[k] = rec[(size_t)i * 32 + 8 + k];
                v.entry_3.hash[k] = rec[(size_t)i * 32 + 16 + k];
                v.entry_4.hash[k] = rec[(size_t)i * 32 + 24 + k];
            }
            v.swap_option.swap = swap[i] != 0;
        }
        return f;
    }
    // PlonkWithPoseidonConstraintSystem::check_poseidon_invocations, values half (plonk_with_poseidon.rs:468-519):
    // permute(swap ? r2 || r1 : r1 || r2) == r3 || r4 for every invocation
    void check_poseidon_invocations() const {
        const size_t n = invocations.size();
        std::vector<uint32_t> l(8 * n), r(8 * n), rate(8 * n), cap(8 * n);
        std::vector<uint8_t> sw(n);
        for (size_t i = 0; i < n; i++) {
            for (int k = 0; k < 8; k++) { l[8 * i + k] = invocations[i].entry_1.hash[k]; r[8 * i + k] = invocations[i].entry_2.hash[k]; }
            sw[i] = invocations[i].swap_option.swap ? 1 : 0;
        }
        check(rsv_poseidon2_half_permute(l.data(), r.data(), sw.data(), rate.data(), cap.data(), n, default_device()),
              "rsv_poseidon2_half_permute");
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 8; k++)
                if (rate[8 * i + k] != invocations[i].entry_3.hash[k] || cap[8 * i + k] != invocations[i].entry_4.hash[k])
                    throw VerificationError(RSV_R_PARSE);
    }
    // log size of the Poseidon component a circuit with `multipliers` such verifications is proved with: the flow is
    // padded to a multiple of 16, at least 32 (PlonkWithPoseidonConstraintSystem::pad, plonk_with_poseidon.rs:282-318),
    // six trace rows per invocation (components/recursive/composition/src/poseidon.rs: first / full / full / partial /
    // full / full), next power of two.  This is the second header word of the next level's proof.
    uint32_t log_size_poseidon(size_t multipliers = 1) const {
        size_t len = invocations.size() * multipliers;
        size_t padded = (len + 15) / 16 * 16;
        if (padded < 32) padded = 32;
        uint32_t lg = 0;
        while (((size_t)1 << lg) < 6 * padded) lg++;
        return lg;
    }
};

// The circuit's `variables: Vec<QM31>` (constraint_system/src/plonk_with_poseidon.rs:19) for batches of proofs of one
// shape: a witness program (include/rsv.h) built from a template proof — the library runs its mirror of the gadgets over
// it once — or loaded from a file, and evaluated on the GPU.  variables(proofs)[i] is what
// PlonkWithPoseidonConstraintSystem holds after the loop body of examples/multi-proofs/src/main.rs:66-139 ran `copies`
// times on proof i.
struct WitnessProgram {
    rsv_witness_program* handle = nullptr;
    rsv_witness_shape shape{};
    uint32_t n_vars = 0;

    WitnessProgram() = default;
    WitnessProgram(const WitnessProgram&) = delete;
    WitnessProgram& operator=(const WitnessProgram&) = delete;
    WitnessProgram(WitnessProgram&& o) noexcept : handle(o.handle), shape(o.shape), n_vars(o.n_vars) { o.handle = nullptr; }
    ~WitnessProgram() { if (handle) rsv_witness_program_destroy(handle); }

    // PoseidonEntry::wire of r1..r4 and SwapOption::addr of every invocation of every copy (shape constants)
    std::vector<std::array<uint32_t, 5>> flow_wires;

    // `multipliers` copies of the verifier of proofs shaped like `template_proof` (which must verify under config / inputs)
    static WitnessProgram build(const std::vector<uint8_t>& template_proof, const PcsConfig& config, const Inputs& inputs, uint32_t multipliers = 1,
                                const std::vector<uint8_t>& set_walks = {}) {
        WitnessProgram p;
        const rsv_pcs_config abi_cfg = config.abi();
        auto pi = abi_inputs(inputs);
        if (!set_walks.empty() && set_walks.size() != multipliers) throw std::runtime_error("set_walks: one entry per copy");
        check(rsv_witness_program_build(template_proof.data(), template_proof.size(), &abi_cfg, pi.data(), pi.size(), multipliers,
                                        set_walks.empty() ? nullptr : set_walks.data(), default_device(), &p.handle), "rsv_witness_program_build");
        check(rsv_witness_program_info(p.handle, &p.n_vars, nullptr, &p.shape), "rsv_witness_program_info");
        p.flow_wires.resize((size_t)p.shape.copies * p.shape.flow_count);
        check(rsv_witness_program_export(p.handle, nullptr, nullptr, &p.flow_wires[0][0]), "rsv_witness_program_export");
        return p;
    }
    // the circuit's Plonk rows (a_wire, b_wire, c_wire, op, poseidon_wire, enforce_c_m31) for the proof whose `variables`
    // are given (empty: the template's): built programs only
    std::vector<std::array<uint32_t, 6>> gates(const std::vector<QM31>& variables = {}) const {
        uint32_t n_rows = 0, n_ops = 0;
        check(rsv_witness_program_gates(handle, &n_rows, &n_ops, nullptr, nullptr), "rsv_witness_program_gates");
        std::vector<std::array<uint32_t, 6>> rows(n_rows);
        std::vector<std::array<uint32_t, 3>> ops(n_ops);
        check(rsv_witness_program_gates(handle, nullptr, nullptr, &rows[0][0], n_ops ? &ops[0][0] : nullptr), "rsv_witness_program_gates");
        if (!variables.empty())
            for (const auto& o : ops) rows[o[0]][3] = variables[o[1]][0] ? o[2] : 0u;  // CirclePointM31Var::select's op follows the bit
        return rows;
    }
    void save(const std::string& path) const {
        uint32_t n_levels = 0;
        check(rsv_witness_program_info(handle, nullptr, &n_levels, nullptr), "rsv_witness_program_info");
        std::vector<uint32_t> instr((size_t)n_vars * 8), levels(n_levels + 1);
        check(rsv_witness_program_export(handle, instr.data(), levels.data(), nullptr), "rsv_witness_program_export");
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write witness program " + path);
        const uint32_t head[4] = {0x57565352u, 1, n_vars, n_levels};
        std::fwrite(head, 4, 4, f);
        std::fwrite(&shape, 4, 9, f);
        std::fwrite(levels.data(), 4, levels.size(), f);
        std::fwrite(instr.data(), 4, instr.size(), f);
        std::fwrite(flow_wires.data(), 4, flow_wires.size() * 5, f);
        std::fclose(f);
    }

    // file = "RSVW" | version 1 | n_vars | n_levels | rsv_witness_shape (9 words) | level_offsets[n_levels + 1] | instr[n_vars][8]
    //        | flow_wires[copies * flow_count][5]
    static WitnessProgram load(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open witness program " + path);
        std::vector<uint32_t> w;
        uint32_t buf[4096];
        size_t got;
        while ((got = std::fread(buf, 4, 4096, f)) > 0) w.insert(w.end(), buf, buf + got);
        std::fclose(f);
        if (w.size() < 13 || w[0] != 0x57565352u || w[1] != 1) throw std::runtime_error("not a witness program: " + path);
        const uint32_t n_vars = w[2], n_levels = w[3];
        WitnessProgram p;
        std::memcpy(&p.shape, &w[4], sizeof(rsv_witness_shape));
        const size_t n_flow = (size_t)p.shape.copies * p.shape.flow_count, at = 13 + (size_t)n_levels + 1 + (size_t)n_vars * 8;
        if (w.size() != at + 5 * n_flow) throw std::runtime_error("truncated witness program: " + path);
        p.flow_wires.resize(n_flow);
        std::memcpy(p.flow_wires.data(), &w[at], 20 * n_flow);
        p.n_vars = n_vars;
        check(rsv_witness_program_create(&w[13 + n_levels + 1], n_vars, &w[13], n_levels, n_vars, &p.shape, default_device(), &p.handle),
              "rsv_witness_program_create");
        return p;
    }
    PcsConfig config() const { return {shape.pow_bits, FriConfig::make(shape.log_last, shape.log_blowup, shape.n_queries)}; }

    // one vector per proof; accept[i] = verified and of the program's shape (the others' vectors are left empty)
    std::vector<std::vector<QM31>> variables(const std::vector<std::vector<uint8_t>>& proofs, const Inputs& inputs,
                                             std::vector<uint8_t>& accept, std::vector<uint8_t>& reason) const {
        const size_t n = proofs.size();
        std::vector<uint64_t> offsets(n + 1, 0);
        for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + proofs[i].size();
        std::vector<uint8_t> blob(offsets[n]);
        for (size_t i = 0; i < n; i++) std::memcpy(blob.data() + offsets[i], proofs[i].data(), proofs[i].size());
        std::vector<uint32_t> vars(n * (size_t)n_vars * 4);
        accept.assign(n, 0);
        reason.assign(n, 0);
        const rsv_pcs_config abi_cfg = config().abi();
        const rsv_cfg_set cfg_set{&abi_cfg, 1, nullptr};
        auto pi = abi_inputs(inputs);
        check(rsv_witness_eval(handle, blob.data(), offsets.data(), n, &cfg_set, pi.data(), pi.size(), vars.data(), nullptr, nullptr,
                               accept.data(), reason.data(), default_device()), "rsv_witness_eval");
        std::vector<std::vector<QM31>> out(n);
        for (size_t i = 0; i < n; i++) {
            if (!accept[i]) continue;
            out[i].resize(n_vars);
            for (uint32_t k = 0; k < n_vars; k++)
                for (int c = 0; c < 4; c++) out[i][k][c] = vars[((size_t)i * n_vars + k) * 4 + c];
        }
        return out;
    }
};

struct Hints {
    // FiatShamirHints (values), DecommitHints, FirstLayerHints, InnerLayersHints of one proof
    FiatShamirResults fiat_shamir;
    // DecommitHints (decommit.rs:186-192): precomputed / trace / interaction / composition proofs, one per query
    std::array<std::vector<SinglePathMerkleProof>, 4> decommit;
    // FirstLayerHints (folding.rs:290-294)
    std::vector<SinglePairMerkleProof> first_layer_merkle_proofs;
    std::vector<std::pair<uint32_t, std::vector<QM31>>> folded_evals_by_column;  // (column log size desc., per query)
    // InnerLayersHints (folding.rs:454-457): per inner layer (log size M-1-i): proofs per query; the folded value of
    // a query at that layer is the proof's self column at the leaf level
    std::vector<std::pair<uint32_t, std::vector<SinglePairMerkleProof>>> inner_layers_merkle_proofs;

    // FiatShamirHints::new(&proof, config, &inputs) (components/hints/src/fiat_shamir.rs:69-74): the configuration is
    // the caller's, never the words serialized in the proof
    static Hints compute(const std::vector<uint8_t>& proof, const PcsConfig& config, const Inputs& inputs) {
        const ProofShape sh = ProofShape::of(proof);
        const uint32_t nq = sh.n_queries, M = sh.M, nt = 1 + sh.n_inner;
        std::vector<uint32_t> row(RSV_TRANSCRIPT_WORDS), tsib((size_t)4 * nq * M * 8), tpos(4 * nq), tcols((size_t)4 * nq * 64),
            fsib((size_t)nt * nq * M * 8), fcols((size_t)nt * nq * 24), ffold((size_t)3 * nq * 4);
        rsv_hints_out ho{};
        ho.n_queries = nq; ho.max_log = M; ho.n_inner = sh.n_inner;
        ho.d_transcript = row.data(); ho.d_trace_sib = tsib.data(); ho.d_trace_pos = tpos.data(); ho.d_trace_cols = tcols.data();
        ho.d_fri_sib = fsib.data(); ho.d_fri_cols = fcols.data(); ho.d_fri_folded = ffold.data();
        const uint64_t offsets[2] = {0, proof.size()};
        uint8_t accept = 0, reason = 0;
        auto pi = abi_inputs(inputs);
        const rsv_pcs_config abi_cfg = config.abi();
        const rsv_cfg_set cfg_set{&abi_cfg, 1, nullptr};
        int st = rsv_verify_hints(proof.data(), offsets, 1, &cfg_set, pi.data(), pi.size(), &ho, &accept, &reason, default_device());
        if (st == RSV_E_SIZE) throw VerificationError(RSV_R_PARSE);  // header shape and body disagree
        check(st, "rsv_verify_hints");
        if (!accept) throw VerificationError((rsv_reason)reason);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(proof.data());
        Hints h;
        auto q4 = [](const uint32_t* p) { return QM31{p[0], p[1], p[2], p[3]}; };
        {
            FiatShamirResults& r = h.fiat_shamir;
            r.max_first_layer_column_log_size = row[3];
            r.z = q4(&row[4]); r.alpha = q4(&row[8]); r.random_coeff = q4(&row[12]); r.oods_t = q4(&row[16]);
            r.oods_x = q4(&row[20]); r.oods_y = q4(&row[24]); r.after_sampled_values_random_coeff = q4(&row[28]);
            for (uint32_t i = 0; i < row[1]; i++) r.fri_alphas.push_back(q4(&row[40 + 4 * i]));
            r.raw_queries.assign(row.begin() + 156, row.begin() + 156 + nq);
        }
        for (int t = 0; t < 4; t++) {
            const uint32_t depth = t == 3 ? M : (sh.A > sh.B ? sh.A : sh.B);
            for (uint32_t i = 0; i < nq; i++) {
                SinglePathMerkleProof p;
                p.query = tpos[t * nq + i];
                p.depth = depth;
                for (int k = 0; k < 8; k++) p.root[k] = w[17 + 8 * t + k];
                for (uint32_t k = 0; k < depth; k++) {
                    Hash hh;
                    for (int e = 0; e < 8; e++) hh[e] = tsib[(((size_t)t * nq + i) * M + k) * 8 + e];
                    p.sibling_hashes.push_back(hh);
                }
                p.columns.assign(depth + 1, {});
                const uint32_t* c = &tcols[((size_t)t * nq + i) * 64];
                if (t == 3) p.columns[M].assign(c, c + 8);
                else {
                    // leaf-level columns first, then the lower log size; at equal sizes plonk columns precede poseidon's
                    const uint32_t hi = depth, lo = sh.A < sh.B ? sh.A : sh.B;
                    uint32_t n_hi = (sh.A == hi ? ProofShape::plonk_cols(t) : 0) + (sh.B == hi ? ProofShape::poseidon_cols(t) : 0);
                    p.columns[hi].assign(c, c + n_hi);
                    if (lo != hi) p.columns[lo].assign(c + n_hi, c + n_hi + (sh.A == lo ? ProofShape::plonk_cols(t) : ProofShape::poseidon_cols(t)));
                }
                h.decommit[t].push_back(std::move(p));
            }
        }
        // column log sizes of the first layer, descending
        std::vector<uint32_t> sizes{M};
        if (sh.A == sh.B) sizes.push_back(sh.A);
        else { sizes.push_back(sh.A > sh.B ? sh.A : sh.B); sizes.push_back(sh.A < sh.B ? sh.A : sh.B); }
        auto pair_proof = [&](uint32_t s, uint32_t i, uint32_t depth, const std::vector<uint32_t>& data_levels, const uint32_t* root) {
            SinglePairMerkleProof p;
            p.query = (h.fiat_shamir.raw_queries[i] & ((1u << M) - 1u)) >> (M - depth);
            p.depth = depth;
            for (int k = 0; k < 8; k++) p.root[k] = root[k];
            for (uint32_t k = 0; k + 1 < depth; k++) {
                Hash hh;
                for (int e = 0; e < 8; e++) hh[e] = fsib[(((size_t)s * nq + i) * M + k) * 8 + e];
                p.sibling_hashes.push_back(hh);
            }
            p.self_columns.assign(depth + 1, std::nullopt);
            p.siblings_columns.assign(depth + 1, std::nullopt);
            for (size_t c = 0; c < data_levels.size(); c++) {
                const uint32_t* v = &fcols[(((size_t)s * nq + i) * 3 + c) * 8];
                p.self_columns[data_levels[c]] = q4(v);
                p.siblings_columns[data_levels[c]] = q4(v + 4);
            }
            return p;
        };
        // FRI layer commitments: walk the variable part
        std::vector<const uint32_t*> commitments;
        {
            size_t pos = 895 + 2;
            for (int t = 0; t < 4; t++) { pos += 2 + 8 * (size_t)w[pos] + 2; }
            pos += 2;
            for (int t = 0; t < 4; t++) { pos += 2 + (size_t)w[pos]; }
            pos += 2;
            auto layer = [&](size_t at) { at += 2 + 4 * (size_t)w[at]; at += 2 + 8 * (size_t)w[at] + 2; commitments.push_back(w + at); return at + 8; };
            pos = layer(pos);
            const uint32_t n_inner = w[pos]; pos += 2;
            for (uint32_t i = 0; i < n_inner; i++) pos = layer(pos);
        }
        for (uint32_t i = 0; i < nq; i++) h.first_layer_merkle_proofs.push_back(pair_proof(0, i, M, sizes, commitments[0]));
        for (size_t g = 0; g < sizes.size(); g++) {
            std::vector<QM31> v;
            for (uint32_t i = 0; i < nq; i++) v.push_back(q4(&ffold[((size_t)g * nq + i) * 4]));
            h.folded_evals_by_column.push_back({sizes[g], std::move(v)});
        }
        for (uint32_t l = 0; l < sh.n_inner; l++) {
            const uint32_t depth = M - 1 - l;
            std::vector<SinglePairMerkleProof> v;
            for (uint32_t i = 0; i < nq; i++) v.push_back(pair_proof(1 + l, i, depth, {depth}, commitments[1 + l]));
            h.inner_layers_merkle_proofs.push_back({depth, std::move(v)});
        }
        return h;
    }
};

// One rsv_ctx per host thread (the reference's types are !Send): keeps the HBM workspace and the pinned staging ring
// alive between calls.
inline rsv_ctx* thread_context() {
    struct Holder {
        rsv_ctx* c = nullptr;
        int device = -1;
        ~Holder() { if (c) rsv_ctx_destroy(c); }
    };
    thread_local Holder h;
    if (!h.c || h.device != default_device()) {
        if (h.c) rsv_ctx_destroy(h.c);
        h.c = nullptr;
        check(rsv_ctx_create(default_device(), &h.c), "rsv_ctx_create");
        h.device = default_device();
    }
    return h.c;
}

// The whole stage sequence of examples/single-proof/src/main.rs:48-82 on a batch.
struct Verifier {
    // accept[i] / reason[i] per proof; never throws for a bad proof.  configs: ONE configuration for the whole batch or
    // one per proof (the recursion chain of examples/multi-proofs/src/main.rs:173-295 mixes six); required — a proof
    // whose header carries another configuration is rejected (RSV_R_PARSE), as the reference never reads it from there.
    static void verify_batch(const std::vector<std::vector<uint8_t>>& proofs, const std::vector<PcsConfig>& configs,
                             const Inputs& inputs, std::vector<uint8_t>& accept, std::vector<uint8_t>& reason) {
        Job job(proofs, configs);
        accept.assign(proofs.size(), 0);
        reason.assign(proofs.size(), 0);
        auto pi = abi_inputs(inputs);
        check(rsv_verify_batch_host(thread_context(), job.ptrs.data(), job.lens.data(), proofs.size(), &job.cfg_set, pi.data(),
                                    pi.size(), accept.data(), reason.data()),
              "rsv_verify_batch_host");
    }
    // The same job over several devices from this ONE process — the reference's driver is one process that walks the whole
    // chain (examples/multi-proofs/src/main.rs:198-295): contiguous shards balanced by bytes (rsv_shard_plan), one host thread per device,
    // verdicts + accept bitmap + count assembled on the host (rsv_multi_verify_batch_host; nothing is exchanged between
    // the devices).  `devices` may name a device more than once.  Returns the number of accepted proofs.
    // How a job of these proofs is cut over `world` ranks: contiguous [lo, hi) per rank, balanced by bytes (rsv_shard_plan) —
    // what verify_batch_multi does inside, and what N processes of one GPU each compute for themselves (same lengths, same plan).
    static std::vector<std::pair<size_t, size_t>> shard_plan(const std::vector<std::vector<uint8_t>>& proofs, size_t world) {
        std::vector<uint64_t> lens(proofs.size());
        for (size_t i = 0; i < proofs.size(); i++) lens[i] = proofs[i].size();
        std::vector<size_t> lo(world), hi(world);
        check(rsv_shard_plan(lens.data(), lens.size(), world, lo.data(), hi.data()), "rsv_shard_plan");
        std::vector<std::pair<size_t, size_t>> out(world);
        for (size_t r = 0; r < world; r++) out[r] = {lo[r], hi[r]};
        return out;
    }
    static uint64_t verify_batch_multi(const std::vector<int>& devices, const std::vector<std::vector<uint8_t>>& proofs,
                                       const std::vector<PcsConfig>& configs, const Inputs& inputs, std::vector<uint8_t>& accept,
                                       std::vector<uint8_t>& reason, std::vector<uint32_t>* bitmap = nullptr) {
        Job job(proofs, configs);
        accept.assign(proofs.size(), 0);
        reason.assign(proofs.size(), 0);
        if (bitmap) bitmap->assign((proofs.size() + 31) / 32, 0u);
        auto pi = abi_inputs(inputs);
        rsv_multi* m = nullptr;
        check(rsv_multi_create(devices.data(), devices.size(), &m), "rsv_multi_create");
        uint64_t count = 0;
        const int st = rsv_multi_verify_batch_host(m, job.ptrs.data(), job.lens.data(), proofs.size(), &job.cfg_set, pi.data(), pi.size(),
                                                   accept.data(), reason.data(), bitmap ? bitmap->data() : nullptr, &count);
        rsv_multi_destroy(m);
        check(st, "rsv_multi_verify_batch_host");
        return count;
    }

   private:
    // one buffer per proof, as the reference holds them (the library gathers, uploads and verifies in a pipeline), and the
    // distinct configurations + per-proof index (rsv_cfg_set)
    struct Job {
        std::vector<rsv_pcs_config> table;
        std::vector<uint8_t> cfg_of;
        std::vector<const uint8_t*> ptrs;
        std::vector<uint64_t> lens;
        rsv_cfg_set cfg_set{};
        Job(const std::vector<std::vector<uint8_t>>& proofs, const std::vector<PcsConfig>& configs) {
            if (configs.empty() || (configs.size() != 1 && configs.size() != proofs.size()))
                throw DeviceError("verify_batch: one PcsConfig, or one per proof", RSV_E_SIZE);
            cfg_of.resize(configs.size() == 1 ? 0 : proofs.size());
            for (size_t i = 0; i < configs.size(); i++) {
                const rsv_pcs_config c = configs[i].abi();
                size_t k = 0;
                while (k < table.size() && !(table[k].pow_bits == c.pow_bits && table[k].log_blowup_factor == c.log_blowup_factor &&
                                             table[k].log_last_layer_degree_bound == c.log_last_layer_degree_bound &&
                                             table[k].n_queries == c.n_queries))
                    k++;
                if (k == table.size()) {
                    if (table.size() == RSV_MAX_CFGS) throw DeviceError("verify_batch: too many distinct configurations", RSV_E_SIZE);
                    table.push_back(c);
                }
                if (!cfg_of.empty()) cfg_of[i] = (uint8_t)k;
            }
            cfg_set = rsv_cfg_set{table.data(), (uint32_t)table.size(), cfg_of.empty() ? nullptr : cfg_of.data()};
            for (auto& p : proofs) { ptrs.push_back(p.data()); lens.push_back(p.size()); }
        }
    };

   public:
    // Reference behaviour for one proof: returns on success, "panics" (throws) at the failing stage.
    static void verify(const std::vector<uint8_t>& proof, const PcsConfig& config, const Inputs& inputs) {
        std::vector<uint8_t> a, r;
        verify_batch({proof}, std::vector<PcsConfig>{config}, inputs, a, r);
        if (!a[0]) throw VerificationError((rsv_reason)r[0]);
    }
};

}  // namespace recursive_stwo
