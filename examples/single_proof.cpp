// examples/single_proof.cpp — the verify half of the reference's examples/single-proof/src/main.rs:23-82 on the
// MI355X library: read one serialized proof, derive the hints (one verifying pass on the GPU), check the per-query
// paths the way the reference's hint constructors do, and run the four verifier stages.
//
//   g++ -std=c++17 -O1 -o single_proof examples/single_proof.cpp -Lrecursive-stwo_amd/csrc -lrsv_hip \
//       -Wl,-rpath,$PWD/recursive-stwo_amd/csrc -Wl,-rpath,/opt/rocm/lib
//   ./single_proof tests/golden/proofs/small_proof.bin
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "../recursive-stwo_amd/host/recursive_stwo.hpp"

using namespace recursive_stwo;

int main(int argc, char** argv) {
    const std::string path = argc > 1 ? argv[1] : "tests/golden/proofs/small_proof.bin";
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 2; }
    std::vector<uint8_t> proof((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());

    // examples/single-proof/src/main.rs:28-31,33
    const PcsConfig config{20, FriConfig::make(2, 5, 16)};
    const Inputs inputs = {{1, QM31{1, 0, 0, 0}}};
    try {
        // FiatShamirHints / DecommitHints / FirstLayerHints / InnerLayersHints (:33-41)
        Hints hints = Hints::compute(proof, config, inputs);
        size_t paths = 0;
        for (auto& tree : hints.decommit) for (auto& p : tree) { p.verify(); paths++; }
        for (auto& p : hints.first_layer_merkle_proofs) { p.verify(); paths++; }
        for (auto& layer : hints.inner_layers_merkle_proofs) for (auto& p : layer.second) { p.verify(); paths++; }
        printf("hints: %zu queries, %zu FRI inner layers, %zu per-query Merkle paths re-verified\n",
               hints.fiat_shamir.raw_queries.size(), hints.inner_layers_merkle_proofs.size(), paths);
        // FiatShamirResults -> CompositionCheck -> AnswerResults -> FoldingResults (:48-82)
        Verifier::verify(proof, config, inputs);
        printf("proof accepted\n");
        // cs.pad(); ...; cs.check_poseidon_invocations() (:85-88), values of the Poseidon accelerator only: the flow the
        // circuit would have recorded, from the GPU's verifying pass; the reference prints the same two sizes (:46-83)
        PoseidonFlow flow = PoseidonFlow::compute(proof, config, inputs);
        flow.check_poseidon_invocations();
        printf("Poseidon circuit size: %zu invocations -> log_size_poseidon %u of the next level's proof\n", flow.invocations.size(),
               flow.log_size_poseidon());
        // what the reference hands to the prover next (:90-98): cs.variables — here from the GPU, through the witness
        // program of this proof's shape (the proof itself serves as the template)
        WitnessProgram program = WitnessProgram::build(proof, config, inputs);
        std::vector<uint8_t> accept, reason;
        auto variables = program.variables({proof}, inputs, accept, reason);
        printf("Plonk circuit: %zu variables (the first witnesses: log sizes %u / %u), %zu flow entries with wires\n", variables[0].size(),
               variables[0][4][0], variables[0][5][0], program.flow_wires.size());
    } catch (const VerificationError& e) {
        printf("proof rejected: %s\n", e.what());
        return 1;
    }
    return 0;
}
