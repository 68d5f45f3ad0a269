// examples/multi_proofs.cpp — batch counterpart of the reference's examples/multi-proofs/src/main.rs:40-139: every
// file named on the command line is one serialized proof (public inputs (1,1), (2,i), (3,u), :49-57); all of them
// are verified in one pipelined call that takes the host buffers as they are.
//
//   g++ -std=c++17 -O1 -o multi_proofs examples/multi_proofs.cpp -Lrecursive-stwo_amd/csrc -lrsv_hip \
//       -Wl,-rpath,$PWD/recursive-stwo_amd/csrc -Wl,-rpath,/opt/rocm/lib
//   ./multi_proofs tests/golden/proofs/level*.bin
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "../recursive-stwo_amd/host/recursive_stwo.hpp"

using namespace recursive_stwo;

int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> proofs;
    for (int i = 1; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        proofs.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    const Inputs inputs = {{1, QM31{1, 0, 0, 0}}, {2, QM31{0, 1, 0, 0}}, {3, QM31{0, 0, 1, 0}}};
    std::vector<uint8_t> accept, reason;
    Verifier::verify_batch(proofs, std::nullopt, inputs, accept, reason);  // config: each proof's own header
    int bad = 0;
    for (size_t i = 0; i < proofs.size(); i++) {
        printf("%-40s %s (stage %u)\n", argv[i + 1], accept[i] ? "accepted" : "REJECTED", (unsigned)reason[i]);
        bad += !accept[i];
    }
    return bad ? 1 : 0;
}
