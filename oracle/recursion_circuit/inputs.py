"""TEST INFRASTRUCTURE (oracle): the circuit's inputs from the CPU oracle — the hint structs of components/hints
(DecommitHints, FirstLayerHints, InnerLayersHints) assembled from oracle/rsv_oracle.c's per-path outputs
(rsvo_trace_paths, rsvo_trace_cols, rsvo_fri_paths).  `ob` is tests/oracle_binding (passed in: this package does not import
tests/)."""
from __future__ import annotations

from . import shape


def build_inputs(proof: bytes, ob, inputs=None):
    """-> ProofData with .decommit[t][i], .first_layer[i], .inner_layers[log_size][i]; queries in transcript order."""
    inputs = ob.STANDARD_INPUTS if inputs is None else inputs
    d = shape.parse_proof(proof)
    sib, pos, depth = ob.trace_paths(proof, d.nq, d.M, inputs)
    assert [int(x) for x in depth] == [max(d.A, d.B)] * 3 + [d.M]
    fsib, fcols = ob.fri_paths(proof, d.nq, d.M, 1 + d.n_inner, inputs)
    d.fill_hints(sib, pos, ob.trace_cols(proof, inputs), fsib, fcols)
    return d
