// examples/multi_proofs.cpp — batch counterpart of the reference's examples/multi-proofs/src/main.rs:40-139: every
// file named on the command line is one serialized proof of the recursion chain (public inputs (1,1), (2,i), (3,u),
// :49-57), verified under the configuration the reference's main() names for that level (:173-295; the file name
// `levelK-*.bin` selects it, anything else is verified under standard_config); all of them go through one pipelined
// call that takes the host buffers as they are.
//
//   g++ -std=c++17 -O1 -o multi_proofs examples/multi_proofs.cpp -Lrecursive-stwo_amd/csrc -lrsv_hip \
//       -Wl,-rpath,$PWD/recursive-stwo_amd/csrc -Wl,-rpath,/opt/rocm/lib
//   ./multi_proofs tests/golden/proofs/level*.bin
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>

#include "../recursive-stwo_amd/host/recursive_stwo.hpp"

using namespace recursive_stwo;

// examples/multi-proofs/src/main.rs:173-196
static const PcsConfig standard_config{20, FriConfig::make(8, 5, 16)}, fast_prover_config{20, FriConfig::make(8, 1, 80)},
    fast_prover2_config{20, FriConfig::make(8, 3, 27)}, fast_verifier_config{23, FriConfig::make(8, 7, 11)},
    fast_verifier2_config{20, FriConfig::make(8, 8, 10)}, fast_verifier3_config{28, FriConfig::make(7, 9, 8)};
// the configuration levelK was PROVEN under = the one the next demo_recurse call verifies it with (:198-295)
static PcsConfig config_of_level(int k) {
    switch (k) {
        case 1: case 4: return fast_prover_config;
        case 2: case 5: return fast_prover2_config;
        case 8: case 9: return fast_verifier_config;
        case 10: case 11: return fast_verifier2_config;
        case 12: case 13: case 14: return fast_verifier3_config;
        default: return standard_config;  // recursive_proof_16_15, level3, level6, level7
    }
}

int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> proofs;
    std::vector<PcsConfig> configs;
    for (int i = 1; i < argc; i++) {
        const std::string name = argv[i];
        const size_t at = name.rfind("level");
        configs.push_back(config_of_level(at == std::string::npos ? 0 : atoi(name.c_str() + at + 5)));
        std::ifstream f(argv[i], std::ios::binary);
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        proofs.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    const Inputs inputs = {{1, QM31{1, 0, 0, 0}}, {2, QM31{0, 1, 0, 0}}, {3, QM31{0, 0, 1, 0}}};
    std::vector<uint8_t> accept, reason;
    if (proofs.empty()) return 0;
    Verifier::verify_batch(proofs, configs, inputs, accept, reason);
    int bad = 0;
    for (size_t i = 0; i < proofs.size(); i++) {
        printf("%-40s %s (stage %u)\n", argv[i + 1], accept[i] ? "accepted" : "REJECTED", (unsigned)reason[i]);
        bad += !accept[i];
    }
    return bad ? 1 : 0;
}
