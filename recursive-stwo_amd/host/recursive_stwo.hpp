// recursive_stwo.hpp — C++ host-side mirror of the reference's interface for the verify path.
//
// Rust is not available in this image, so the host layer above the C-ABI (include/rsv.h) is C++.
// Names, argument meaning and failure behaviour follow the reference's Rust items (values only —
// the gate-recording half of every *Var type is out of scope):
//
//   poseidon31::poseidon2_permute          primitives/poseidon31/src/implementation.rs:108-149
//   Poseidon2HalfVar::{permute, permute_get_rate, permute_get_capacity, swap_permute_get_*}
//                                          primitives/poseidon31/src/lib.rs:251-423
//   Poseidon31MerkleHasherVar::*           primitives/merkle/src/lib.rs:9-181
//   ChannelVar::{mix_root, draw_felts, mix_one_felt, mix_two_felts}
//                                          primitives/channel/src/lib.rs:24-58
//   SinglePathMerkleProof::verify          components/hints/src/decommit.rs:22-42
//   FiatShamirResults::compute             components/recursive/fiat_shamir/src/lib.rs:31-176
//   verify (FiatShamir → Composition → Answer → Folding)   examples/single-proof/src/main.rs:48-82
//
// The reference panics (assert_eq!/unwrap, panic = 'abort') when a check fails; here a failed check
// throws recursive_stwo::VerificationError carrying the stage, and API/device failures throw
// recursive_stwo::DeviceError.  Every call runs on the GPU through librsv_hip.so; there is no CPU path.
// These single-item wrappers exist for drop-in parity; throughput comes from Verifier::verify_batch.
#pragma once
#include <array>
#include <cstdint>
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

// The whole stage sequence of examples/single-proof/src/main.rs:48-82 on a batch.
struct Verifier {
    // accept[i] / reason[i] per proof; never throws for a bad proof.
    static void verify_batch(const std::vector<std::vector<uint8_t>>& proofs, const std::optional<PcsConfig>& config,
                             const Inputs& inputs, std::vector<uint8_t>& accept, std::vector<uint8_t>& reason) {
        std::vector<uint8_t> blob;
        std::vector<uint64_t> offsets{0};
        for (auto& p : proofs) { blob.insert(blob.end(), p.begin(), p.end()); offsets.push_back(blob.size()); }
        accept.assign(proofs.size(), 0);
        reason.assign(proofs.size(), 0);
        auto pi = abi_inputs(inputs);
        rsv_pcs_config cfg{};
        if (config) cfg = config->abi();
        check(rsv_verify_batch(blob.data(), offsets.data(), proofs.size(), config ? &cfg : nullptr, pi.data(), pi.size(),
                               accept.data(), reason.data(), default_device()),
              "rsv_verify_batch");
    }
    // Reference behaviour for one proof: returns on success, "panics" (throws) at the failing stage.
    static void verify(const std::vector<uint8_t>& proof, const PcsConfig& config, const Inputs& inputs) {
        std::vector<uint8_t> a, r;
        verify_batch({proof}, config, inputs, a, r);
        if (!a[0]) throw VerificationError((rsv_reason)r[0]);
    }
};

}  // namespace recursive_stwo
