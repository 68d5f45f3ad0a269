"""TEST INFRASTRUCTURE (oracle): what the recursion circuit is fed — the proof as the reference's `*Var` types see it and
the hint structs of components/hints (FiatShamirHints, DecommitHints, FirstLayerHints, InnerLayersHints), assembled
from the C oracle's per-path outputs (oracle/rsv_oracle.h: rsvo_trace_paths, rsvo_trace_cols, rsvo_fri_paths), which are
the same buffers rsv_verify_hints_dev fills on the GPU (include/rsv.h).  `ob` is tests/oracle_binding (passed in: this
package does not import tests/)."""
from __future__ import annotations

import numpy as np

PLONK_COLS = (10, 12, 8)
POSEIDON_COLS = (40, 48, 8)


class ProofData:
    pass


def parse_proof(proof: bytes) -> ProofData:
    """bincode layout: SURVEY App. A."""
    w = np.frombuffer(proof, dtype=np.uint32)
    d = ProofData()
    d.lp, d.lq = int(w[0]), int(w[1])
    d.plonk_total_sum = tuple(int(x) for x in w[2:6])
    d.poseidon_total_sum = tuple(int(x) for x in w[6:10])
    d.pow_bits, d.blowup, d.log_last, d.nq = int(w[10]), int(w[11]), int(w[12]), int(w[13])
    assert int(w[15]) == 4
    d.commitments = [tuple(int(x) for x in w[17 + 8 * t:25 + 8 * t]) for t in range(4)]
    pos = 49
    assert int(w[pos]) == 4
    pos += 2
    d.sampled_values = []
    for t, ncols in enumerate((50, 60, 16, 8)):
        assert int(w[pos]) == ncols
        pos += 2
        tree = []
        for _ in range(ncols):
            ns = int(w[pos]); pos += 2
            col = []
            for _ in range(ns):
                col.append(tuple(int(x) for x in w[pos:pos + 4])); pos += 4
            tree.append(col)
        d.sampled_values.append(tree)
    assert pos == 895
    pos += 2
    for _ in range(4):
        nh = int(w[pos]); pos += 2 + 8 * nh
        pos += 2
    pos += 2
    for _ in range(4):
        nv = int(w[pos]); pos += 2 + nv
    d.nonce = int(w[pos]) | (int(w[pos + 1]) << 32)
    pos += 2

    def layer(pos):
        nw = int(w[pos]); pos += 2 + 4 * nw
        nh = int(w[pos]); pos += 2 + 8 * nh
        pos += 2
        return pos + 8, tuple(int(x) for x in w[pos:pos + 8])

    pos, d.first_layer_commitment = layer(pos)
    n_inner = int(w[pos]); pos += 2
    d.inner_layer_commitments = []
    for _ in range(n_inner):
        pos, c = layer(pos)
        d.inner_layer_commitments.append(c)
    n_last = int(w[pos]); pos += 2
    d.last_poly = [tuple(int(x) for x in w[pos + 4 * k:pos + 4 * k + 4]) for k in range(n_last)]
    pos += 4 * n_last
    assert int(w[pos]) == d.log_last and pos + 1 == len(w)
    d.A, d.B = d.lp + d.blowup, d.lq + d.blowup
    d.M = max(d.lp + 1, d.lq + 2) + d.blowup
    d.n_inner = n_inner
    return d


class PathProof:
    """SinglePathMerkleProof (components/hints/src/decommit.rs:10-19)."""

    def __init__(self, query, sibling_hashes, columns, depth):
        self.query, self.sibling_hashes, self.columns, self.depth = query, sibling_hashes, columns, depth


class PairProof:
    """SinglePairMerkleProof (components/hints/src/folding.rs:20-29)."""

    def __init__(self, query, sibling_hashes, self_columns, siblings_columns, depth):
        self.query, self.sibling_hashes, self.depth = query, sibling_hashes, depth
        self.self_columns, self.siblings_columns = self_columns, siblings_columns


def build_inputs(proof: bytes, ob, inputs=None):
    """-> ProofData with .decommit[t][i] (PathProof), .first_layer[i], .inner_layers[log_size][i] (PairProof), queries
    in transcript (draw) order."""
    inputs = ob.STANDARD_INPUTS if inputs is None else inputs
    d = parse_proof(proof)
    nq, M, A, B = d.nq, d.M, d.A, d.B
    sib, pos, depth = ob.trace_paths(proof, nq, M, inputs)
    cols = ob.trace_cols(proof, inputs)
    d.decommit = []
    for t in range(4):
        dep = int(depth[t])
        assert dep == (M if t == 3 else max(A, B))
        if t == 3:
            levels = [(M, 8)]
        elif A == B:
            levels = [(A, PLONK_COLS[t] + POSEIDON_COLS[t])]
        else:  # leaf level first, as rsvo_trace_cols packs them
            levels = sorted([(A, PLONK_COLS[t]), (B, POSEIDON_COLS[t])], reverse=True)
        proofs = []
        for i in range(nq):
            columns, off = {}, 0
            for log_size, n in levels:
                columns[log_size] = [int(x) for x in cols[t, i, off:off + n]]
                off += n
            proofs.append(PathProof(int(pos[t, i]), [tuple(int(x) for x in sib[t, i, k]) for k in range(dep)], columns, dep))
        d.decommit.append(proofs)
    fsib, fcols = ob.fri_paths(proof, nq, M, 1 + d.n_inner, inputs)
    data_levels = sorted({M, A, B}, reverse=True)
    queries_M = [int(x) for x in pos[3]]
    d.first_layer = []
    for i in range(nq):
        selfc = {ls: tuple(int(x) for x in fcols[0, i, c, 0:4]) for c, ls in enumerate(data_levels)}
        sibc = {ls: tuple(int(x) for x in fcols[0, i, c, 4:8]) for c, ls in enumerate(data_levels)}
        d.first_layer.append(PairProof(queries_M[i], [tuple(int(x) for x in fsib[0, i, k]) for k in range(M - 1)], selfc, sibc, M))
    d.inner_layers = {}
    for l in range(d.n_inner):
        ls = M - 1 - l
        proofs = []
        for i in range(nq):
            selfc = {ls: tuple(int(x) for x in fcols[1 + l, i, 0, 0:4])}
            sibc = {ls: tuple(int(x) for x in fcols[1 + l, i, 0, 4:8])}
            proofs.append(PairProof(queries_M[i] >> (M - ls), [tuple(int(x) for x in fsib[1 + l, i, k]) for k in range(ls - 1)],
                                    selfc, sibc, ls))
        d.inner_layers[ls] = proofs
    d.queries_M = queries_M
    return d
