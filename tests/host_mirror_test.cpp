// host_mirror_test.cpp — the reference's own unit tests, restated against the C++ host mirror
// (recursive-stwo_amd/host/recursive_stwo.hpp).  Runs on the GPU box (tests/test_host_mirror.py).
//   test_poseidon2_permute      primitives/poseidon31/src/implementation.rs:157-172
//   test_consistency (merkle)   primitives/merkle/src/lib.rs:206-303 (expected values: SURVEY App. C)
//   test_fiat_shamir            components/recursive/fiat_shamir/src/lib.rs:197-236 (challenges: SURVEY App. C)
//   test_folding (= full verify) components/recursive/folding/src/lib.rs:231-303
//   test_poseidon_flow          constraint_system/src/plonk_with_poseidon.rs:282-318,468-519 on the GPU-recorded flow
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../recursive-stwo_amd/host/recursive_stwo.hpp"

using namespace recursive_stwo;

#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } \
    } while (0)

static std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    EXPECT(f.good());
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void test_poseidon2_permute() {
    std::array<M31, 16> state{};
    for (uint32_t i = 0; i < 16; i++) state[i] = i;
    poseidon31::poseidon2_permute(state);
    const std::array<M31, 16> want = {260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083,
                                      755063943, 1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267,
                                      1269806256, 1287849264};
    EXPECT(state == want);
}

// primitives/poseidon31/src/emulated.rs:236-275: the three permutes of the reference's test through the gate-level
// form; the returned halves are the known answer, 401 rows (+12 with a swap bit) are appended per call.
static void test_poseidon2_emulated_permute() {
    const std::array<M31, 16> want = {260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083,
                                      755063943, 1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267,
                                      1269806256, 1287849264};
    const std::array<QM31, 2> lo = {QM31{0, 1, 2, 3}, QM31{4, 5, 6, 7}}, hi = {QM31{8, 9, 10, 11}, QM31{12, 13, 14, 15}};
    using poseidon31::emulated::IsSwap;
    auto res = poseidon31::emulated::poseidon_permute_emulated_batch({lo, lo, hi}, {hi, hi, lo}, {IsSwap{}, IsSwap{false}, IsSwap{true}});
    EXPECT(res.size() == 3);
    for (size_t p = 0; p < 3; p++) {
        std::array<M31, 16> got{};
        for (int i = 0; i < 4; i++) {
            got[i] = res[p].out_left[0][i]; got[4 + i] = res[p].out_left[1][i];
            got[8 + i] = res[p].out_right[0][i]; got[12 + i] = res[p].out_right[1][i];
        }
        EXPECT(got == want);
        EXPECT(res[p].rows.size() == 401);
        EXPECT(res[p].swap_rows.size() == (p == 0 ? 0u : 12u));
    }
    // Some((true, _)) exchanges the halves: the last swap row, new_right_1 = right_1 - (right_1 - left_1), is left_1
    EXPECT((res[2].swap_rows[11] == QM31{12, 13, 14, 15}));
    auto one = poseidon31::emulated::poseidon_permute_emulated(lo, hi, std::nullopt);
    EXPECT(one.rows == res[0].rows);
}

static void test_merkle_consistency() {
    // hash_m31_columns_get_rate == Poseidon31MerkleHasher::hash_node(None, cols)
    for (size_t len : {7u, 13u, 16u, 17u, 21u, 25u}) {
        std::vector<M31> cols(len);
        for (size_t i = 0; i < len; i++) cols[i] = (uint32_t)(1000003u * (i + 1) + len) % RSV_M31_P;
        HashVar a = Poseidon31MerkleHasherVar::hash_m31_columns_get_rate(cols);
        HashVar b = HashVar::permute_get_rate(HashVar::zero(), Poseidon31MerkleHasherVar::hash_m31_columns_get_capacity(cols));
        EXPECT(a.value == b.value);
    }
    EXPECT((Poseidon31MerkleHasherVar::hash_m31_columns_get_rate({1, 2, 3, 4, 5}).value ==
            Hash{557709851, 1113733662, 222169927, 1376019790, 387901840, 1087892516, 628125718, 969660801}));
    M31 l8[8] = {1, 2, 3, 4, 5, 6, 7, 8}, r8[8] = {9, 10, 11, 12, 13, 14, 15, 16};
    HashVar l = HashVar::from_m31(l8), r = HashVar::from_m31(r8);
    EXPECT((Poseidon31MerkleHasherVar::hash_tree(l, r).value ==
            Hash{164793487, 387042994, 621688597, 428853092, 1214488792, 1623406829, 1918424220, 1537261691}));
    HashVar col = Poseidon31MerkleHasherVar::hash_m31_columns_get_capacity({7, 7, 7});
    EXPECT((Poseidon31MerkleHasherVar::hash_tree_with_column(l, r, col).value ==
            Hash{2077916493, 57586551, 1709117860, 800174306, 352135528, 1590574078, 1798659285, 1176940757}));
    // swap variants: primitives/merkle/src/lib.rs:22-41
    EXPECT(Poseidon31MerkleHasherVar::hash_tree_with_swap(l, r, true).value == Poseidon31MerkleHasherVar::hash_tree(r, l).value);
}

static void test_channel(const std::vector<uint8_t>& small) {
    // commitment[0] of small_proof.bin lives at byte 68; digest after mix_root: SURVEY App. C
    HashVar c0;
    for (int i = 0; i < 8; i++) {
        uint32_t w;
        memcpy(&w, small.data() + 68 + 4 * i, 4);
        c0.value[i] = w;
    }
    EXPECT((c0.value == Hash{408034276, 2096354985, 1338871816, 1690865784, 2073546231, 904203018, 917926113, 1884771894}));
    ChannelVar channel;
    channel.mix_root(c0);
    EXPECT((channel.digest.value == Hash{1662218662, 1343911353, 1635652531, 1581605795, 624037075, 24422224, 442561826, 420390772}));
    EXPECT(channel.n_sent == 0);
    auto d0 = channel.draw_felts();
    auto d1 = channel.draw_felts();
    EXPECT(channel.n_sent == 2 && d0 != d1);
}

static void test_fiat_shamir(const std::vector<uint8_t>& small) {
    FiatShamirResults r = FiatShamirResults::compute(small);
    EXPECT((r.z == QM31{1211683141, 437669427, 409200369, 1127771350}));
    EXPECT((r.alpha == QM31{608237629, 60905622, 1129253272, 1937554417}));
    EXPECT((r.random_coeff == QM31{510535785, 709795745, 2021304333, 468388088}));
    EXPECT((r.oods_t == QM31{432538781, 1881392761, 1838851372, 291147612}));
    EXPECT((r.after_sampled_values_random_coeff == QM31{258757294, 1276317760, 1536227746, 6873968}));
    EXPECT(r.fri_alphas.size() == 8);
    EXPECT((r.fri_alphas[0] == QM31{2118644044, 1562770230, 410003546, 1078681992}));
    EXPECT((r.fri_alphas[7] == QM31{1686932136, 243723819, 74374586, 128365204}));
    EXPECT(r.raw_queries.size() == 16 && r.max_first_layer_column_log_size == 15);
    EXPECT((r.raw_queries[0] & 0x7fff) == 3311 && (r.raw_queries[15] & 0x7fff) == 23721);
}

static void test_verify(const std::vector<uint8_t>& small) {
    PcsConfig config{20, FriConfig::make(2, 5, 16)};
    Inputs inputs = {{1, QM31{1, 0, 0, 0}}};
    Verifier::verify(small, config, inputs);  // must not throw
    bool threw = false;
    try {
        Verifier::verify(small, PcsConfig{20, FriConfig::make(2, 5, 15)}, inputs);
    } catch (const VerificationError& e) { threw = e.reason == RSV_R_PARSE; }
    EXPECT(threw);
    threw = false;
    try {
        Verifier::verify(small, config, {{1, QM31{2, 0, 0, 0}}});
    } catch (const VerificationError& e) { threw = e.reason == RSV_R_LOGUP; }
    EXPECT(threw);
    auto bad = small;
    bad[30000] ^= 1;
    std::vector<uint8_t> acc, reason;
    Verifier::verify_batch({small, bad}, {config}, inputs, acc, reason);
    EXPECT(acc[0] == 1 && acc[1] == 0 && reason[1] != 0);
}

// components/hints/src/decommit.rs test_decommitment + folding.rs test_folding: derive the hints of small_proof.bin
// and verify every per-query path against its commitment.
static void test_hints(const std::vector<uint8_t>& small) {
    Inputs inputs = {{1, QM31{1, 0, 0, 0}}};
    const PcsConfig config{20, FriConfig::make(2, 5, 16)};
    Hints h = Hints::compute(small, config, inputs);
    EXPECT((h.fiat_shamir.z == QM31{1211683141, 437669427, 409200369, 1127771350}));
    EXPECT(h.fiat_shamir.raw_queries.size() == 16);
    for (int t = 0; t < 4; t++) {
        EXPECT(h.decommit[t].size() == 16);
        for (const auto& p : h.decommit[t]) p.verify();  // throws on a root mismatch
    }
    EXPECT(h.first_layer_merkle_proofs.size() == 16);
    for (const auto& p : h.first_layer_merkle_proofs) p.verify();
    EXPECT(h.folded_evals_by_column.size() == 3 && h.folded_evals_by_column[0].first == 15);
    EXPECT(h.inner_layers_merkle_proofs.size() == 7);
    for (const auto& layer : h.inner_layers_merkle_proofs) {
        EXPECT(layer.second.size() == 16 && layer.second[0].depth == layer.first);
        for (size_t i = 0; i < layer.second.size(); i += 5) layer.second[i].verify();
    }
    // a tampered sibling hash must fail the path it belongs to
    auto bad = h.decommit[2][3];
    bad.sibling_hashes[4][0] ^= 1;
    bool threw = false;
    try { bad.verify(); } catch (const VerificationError&) { threw = true; }
    EXPECT(threw);
    auto badp = h.first_layer_merkle_proofs[1];
    (*badp.siblings_columns[15])[2] ^= 1;
    threw = false;
    try { badp.verify(); } catch (const VerificationError&) { threw = true; }
    EXPECT(threw);
    // wrong public input: the hints constructor "panics" at the logup check
    threw = false;
    try { Hints::compute(small, config, {{1, QM31{2, 0, 0, 0}}}); } catch (const VerificationError& e) { threw = e.reason == RSV_R_LOGUP; }
    EXPECT(threw);
}

// constraint_system/src/plonk_with_poseidon.rs:468-519 (check_poseidon_invocations) + :282-318 (pad) on the flow the
// GPU's verifying pass records: every invocation permutes to its outputs, the transcript part is the channel chain, and
// the padded flow of small_proof.bin's verification is the 2^15-row Poseidon component of recursive_proof_16_15.bin
// (examples/single-proof/src/main.rs:100-103), that proof's verified 5 times level1-5.bin's 2^18 rows
// (examples/multi-proofs/src/main.rs:198-204).
static void test_poseidon_flow(const std::vector<uint8_t>& small, const std::vector<uint8_t>& rec, const std::vector<uint8_t>& level1) {
    PoseidonFlow f = PoseidonFlow::compute(small, PcsConfig{20, FriConfig::make(2, 5, 16)}, {{1, QM31{1, 0, 0, 0}}});
    EXPECT(f.invocations.size() == 3481);
    f.check_poseidon_invocations();
    // mix_root(commitment 0) on the zero digest, then mix_one_felt(log_size_plonk) on its capacity
    const uint32_t* w = reinterpret_cast<const uint32_t*>(small.data());
    for (int k = 0; k < 8; k++) EXPECT(f.invocations[0].entry_1.hash[k] == w[17 + k] && f.invocations[0].entry_2.hash[k] == 0);
    EXPECT(f.invocations[1].entry_2.hash == f.invocations[0].entry_4.hash && f.invocations[1].entry_1.hash[0] == w[0]);
    EXPECT(f.log_size_poseidon() == reinterpret_cast<const uint32_t*>(rec.data())[1] && f.log_size_poseidon() == 15);
    Inputs std_inputs = {{1, QM31{1, 0, 0, 0}}, {2, QM31{0, 1, 0, 0}}, {3, QM31{0, 0, 1, 0}}};
    PoseidonFlow g = PoseidonFlow::compute(rec, PcsConfig{20, FriConfig::make(8, 5, 16)}, std_inputs);
    EXPECT(g.invocations.size() == 5289);
    g.check_poseidon_invocations();
    EXPECT(g.log_size_poseidon(5) == reinterpret_cast<const uint32_t*>(level1.data())[1] && g.log_size_poseidon(5) == 18);
    // a flipped output word is caught
    g.invocations[1234].entry_3.hash[3] ^= 1;
    bool threw = false;
    try { g.check_poseidon_invocations(); } catch (const VerificationError&) { threw = true; }
    EXPECT(threw);
}

// The circuit's `variables` for a batch: the program file is written by the Python side of the host layer
// (WitnessProgram.build(...).export().save_raw) for the shape of level10-1.bin; level11-1.bin is another proof of that
// shape, level12-1.bin is not.  What the reference's check_arithmetics looks at first — the four fixed variables 0, 1, i, j
// (plonk_with_poseidon.rs:61-64) — and the statement words the proof allocation pushes next (data_structures/src/lib.rs:36-46).
static void test_witness(const std::string& program_path, const std::vector<uint8_t>& l10, const std::vector<uint8_t>& l11,
                         const std::vector<uint8_t>& l12) {
    WitnessProgram prog = WitnessProgram::load(program_path);
    EXPECT(prog.shape.log_size_plonk == 16 && prog.shape.log_size_poseidon == 15 && prog.shape.n_queries == 10 && prog.shape.copies == 1);
    EXPECT(prog.flow_wires.size() == prog.shape.flow_count && prog.flow_wires[0][0] != 0 && prog.flow_wires[0][4] == 0);
    Inputs inputs = {{1, QM31{1, 0, 0, 0}}, {2, QM31{0, 1, 0, 0}}, {3, QM31{0, 0, 1, 0}}};
    std::vector<uint8_t> accept, reason;
    auto tampered = l10;
    tampered[60000] ^= 1;
    auto vars = prog.variables({l10, l11, tampered, l12}, inputs, accept, reason);
    EXPECT(accept == (std::vector<uint8_t>{1, 1, 0, 0}) && reason[2] != 0);
    EXPECT(vars[0].size() == prog.n_vars && vars[1].size() == prog.n_vars && vars[2].empty() && vars[3].empty());
    for (int k = 0; k < 2; k++) {
        EXPECT((vars[k][0] == QM31{0, 0, 0, 0}) && (vars[k][1] == QM31{1, 0, 0, 0}) && (vars[k][2] == QM31{0, 1, 0, 0}) &&
               (vars[k][3] == QM31{0, 0, 1, 0}));
    }
    EXPECT(vars[0] != vars[1]);
    // the first witnesses are the statement: log sizes, then the two claimed sums as they stand in the proof
    for (int k = 0; k < 2; k++) {
        const auto& proof = k == 0 ? l10 : l11;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(proof.data());
        size_t at = 4;
        while (at < vars[k].size() && !(vars[k][at] == QM31{16, 0, 0, 0} && vars[k][at + 1] == QM31{15, 0, 0, 0})) at++;
        EXPECT(at + 3 < vars[k].size());
        EXPECT((vars[k][at + 2] == QM31{w[2], w[3], w[4], w[5]}) && (vars[k][at + 3] == QM31{w[6], w[7], w[8], w[9]}));
    }
    printf("witness: %u variables per proof, accept = %d %d %d %d\n", prog.n_vars, accept[0], accept[1], accept[2], accept[3]);
    // built here from level11-1.bin as the template: the same program (it depends on the shape only), so the same vectors;
    // and it survives a round trip through a file
    WitnessProgram built = WitnessProgram::build(l11, prog.config(), inputs);
    EXPECT(built.n_vars == prog.n_vars && built.flow_wires == prog.flow_wires);
    std::vector<uint8_t> a2, r2;
    auto vars2 = built.variables({l10, l11}, inputs, a2, r2);
    EXPECT(vars2[0] == vars[0] && vars2[1] == vars[1]);
    // the gate list for each of the two proofs: the same wires, `op` differing where it follows the witness; every gate holds
    const auto g10 = built.gates(vars[0]), g11 = built.gates(vars[1]);
    EXPECT(g10.size() == g11.size() && g10 != g11);
    const uint64_t P = 0x7fffffffull;
    for (size_t i = 0; i < g11.size(); i++) {
        EXPECT(g10[i][0] == g11[i][0] && g10[i][1] == g11[i][1] && g10[i][2] == g11[i][2] && g10[i][4] == g11[i][4] && g10[i][5] == g11[i][5]);
        const auto &a = vars[1][g11[i][0]], &b = vars[1][g11[i][1]], &c = vars[1][g11[i][2]];
        const uint64_t op = g11[i][3];
        if (op == 1) { for (int k = 0; k < 4; k++) EXPECT((a[k] + (uint64_t)b[k]) % P == c[k]); }
        else if (op > 1) { for (int k = 0; k < 4; k++) EXPECT(op * ((a[k] + (uint64_t)b[k]) % P) % P == c[k]); }
    }
    built.save(program_path + ".copy");
    WitnessProgram again = WitnessProgram::load(program_path + ".copy");
    auto vars3 = again.variables({l11}, inputs, a2, r2);
    EXPECT(a2[0] == 1 && vars3[0] == vars[1]);
}

// SURVEY 8e through the C-ABI from ONE process (the reference's driver is one process, examples/multi-proofs/src/main.rs:198-295):
// three contexts on device 0 stand in for three GPUs; 10 proofs -> shards of 4 / 3 / 3 under three configurations; the
// verdicts must be the single-context ones, the bitmap bit i = accept[i], the count their sum.
static void test_multi(const std::vector<uint8_t>& small, const std::vector<uint8_t>& rec, const std::vector<uint8_t>& level1) {
    const PcsConfig c_small{20, FriConfig::make(2, 5, 16)}, c_std{20, FriConfig::make(8, 5, 16)}, c_fast{20, FriConfig::make(8, 1, 80)};
    const Inputs inputs = {{1, QM31{1, 0, 0, 0}}, {2, QM31{0, 1, 0, 0}}, {3, QM31{0, 0, 1, 0}}};
    std::vector<std::vector<uint8_t>> proofs;
    std::vector<PcsConfig> configs;
    for (int i = 0; i < 10; i++) {
        const int k = i % 3;
        proofs.push_back(k == 0 ? rec : (k == 1 ? level1 : small));  // small_proof.bin has ONE public input: rejected under three
        configs.push_back(k == 0 ? c_std : (k == 1 ? c_fast : c_small));
        if (i == 3 || i == 4) proofs.back()[proofs.back().size() / 2 + 64 * i] ^= 4;
    }
    proofs[9].clear();  // an empty buffer is a malformed proof, not an error
    std::vector<uint8_t> a1, r1, a3, r3;
    Verifier::verify_batch(proofs, configs, inputs, a1, r1);
    std::vector<uint32_t> bitmap;
    const uint64_t count = Verifier::verify_batch_multi({0, 0, 0}, proofs, configs, inputs, a3, r3, &bitmap);
    EXPECT(a3 == a1 && r3 == r1);
    EXPECT(a1[0] == 1 && a1[1] == 1 && a1[3] == 0 && a1[4] == 0 && a1[6] == 1 && a1[7] == 1 && a1[9] == 0 && r1[9] == RSV_R_PARSE);
    uint64_t want = 0;
    for (size_t i = 0; i < a3.size(); i++) {
        want += a3[i];
        EXPECT(((bitmap[i >> 5] >> (i & 31)) & 1u) == a3[i]);
    }
    EXPECT(count == want && bitmap.size() == 1 && (bitmap[0] >> 10) == 0);
    size_t lo, hi;
    rsv_shard_range(10, 0, 3, &lo, &hi);
    EXPECT(lo == 0 && hi == 4);
    rsv_shard_range(10, 2, 3, &lo, &hi);
    EXPECT(lo == 7 && hi == 10);
    // the byte-balanced plan over this job (proof 9 is empty): contiguous, covering, and the heaviest shard within one proof of the lightest
    {
        const auto plan = Verifier::shard_plan(proofs, 3);
        EXPECT(plan.size() == 3 && plan[0].first == 0 && plan[2].second == proofs.size() && plan[0].second == plan[1].first && plan[1].second == plan[2].first);
        size_t bytes[3] = {0, 0, 0}, longest = 0;
        for (size_t r = 0; r < 3; r++)
            for (size_t i = plan[r].first; i < plan[r].second; i++) { bytes[r] += proofs[i].size(); longest = std::max(longest, proofs[i].size()); }
        const size_t mx = std::max(bytes[0], std::max(bytes[1], bytes[2])), mn = std::min(bytes[0], std::min(bytes[1], bytes[2]));
        EXPECT(mx - mn <= 2 * longest);
        rsv_pcs_config beyond{20, 5, 8, 129};
        EXPECT(rsv_cfg_check(&beyond) == RSV_E_SIZE);
    }
    // more contexts than proofs: the empty shards are skipped
    std::vector<std::vector<uint8_t>> two(proofs.begin(), proofs.begin() + 2);
    std::vector<PcsConfig> two_cfg(configs.begin(), configs.begin() + 2);
    EXPECT(Verifier::verify_batch_multi({0, 0, 0, 0, 0}, two, two_cfg, inputs, a3, r3) == 2);
    printf("multi: 3 contexts, accept =");
    for (auto a : a1) printf(" %d", a);
    printf(", count = %llu\n", (unsigned long long)count);
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "tests/golden/proofs";
    auto small = read_file(dir + "/small_proof.bin");
    test_poseidon2_permute();
    test_poseidon2_emulated_permute();
    test_merkle_consistency();
    test_channel(small);
    test_fiat_shamir(small);
    test_verify(small);
    test_hints(small);
    test_poseidon_flow(small, read_file(dir + "/recursive_proof_16_15.bin"), read_file(dir + "/level1-5.bin"));
    test_multi(small, read_file(dir + "/recursive_proof_16_15.bin"), read_file(dir + "/level1-5.bin"));
    if (argc > 2) test_witness(argv[2], read_file(dir + "/level10-1.bin"), read_file(dir + "/level11-1.bin"), read_file(dir + "/level12-1.bin"));
    printf("host mirror: all tests passed\n");
    return 0;
}
