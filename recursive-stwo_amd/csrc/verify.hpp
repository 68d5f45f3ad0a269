// verify.hpp — the batch verification pipeline for Poseidon31-channel
// Plonk-with-Poseidon Circle-STARK proofs, as HIP kernels for gfx950.
//
// Stage            kernel (file)                      parallelism                  reference
// ---------------  ---------------------------------  ---------------------------  ----------------------------------------
// wire format      k_parse (k_parse.hpp)              1 lane / proof               bincode of PlonkWithPoseidonProof
// canonicity       (in the kernel that reads a word)  —                            M31 words must be < P: layout.hpp; fallback in k_finalize
// transcript       k_transcript[_row] (k_transcript)  1 lane or 1 DPP row / proof  components/recursive/fiat_shamir/src/lib.rs:31-176
// OODS identity    k_oods (k_oods.hpp)                1 lane / proof               components/recursive/composition/src/**
// decommit plan    k_plan_par (k_plan.hpp)            1 lane / (proof, query)      components/hints/src/decommit.rs:53-142, folding.rs:107-212
// quotient consts  k_qconst (k_plan.hpp)              1 lane / proof               components/recursive/answer/src/data_structures.rs:132-189
// quotients+folds  k_query<QB> (k_query.hpp)          1 lane / (proof, query)      components/recursive/answer/src/**, folding/src/lib.rs:57-204
// column hashing   k_row_hash (k_merkle.hpp)          1 lane / (proof, tree, row)  primitives/merkle/src/lib.rs:50-181
// trace trees      k_trace_merkle (k_merkle.hpp)      1 lane / (proof,tree,query)  components/recursive/data_structures/src/lib.rs:315-354
// FRI trees        k_pair_merkle (k_merkle.hpp)       1 lane / (proof,layer,query) components/recursive/data_structures/src/lib.rs:400-464
// verdict          k_finalize (k_finalize.hpp)        1 lane / proof               accept bit + first failing stage
//
// The Merkle kernels follow the reference's per-query form: every lane walks
// one authentication path from its leaf to the root, one Poseidon2 permutation
// per level.  Sibling hashes that the reference's host code re-derives from
// other query paths (SinglePathMerkleProof::from_stwo_proof) are exchanged
// between the lanes of a proof through LDS; the remaining ones are gathered
// straight from the proof's hash_witness at the index given by the plan.
#pragma once
#include "verify_common.hpp"
#include "k_parse.hpp"
#include "k_transcript.hpp"
#include "k_oods.hpp"
#include "k_plan.hpp"
#include "k_query.hpp"
#include "k_merkle.hpp"
#include "k_finalize.hpp"
