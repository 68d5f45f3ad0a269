"""TEST INFRASTRUCTURE (oracle): what the circuit is fed: the proof as the reference's `*Var` types see it and the per-query hint structs of
components/hints — SinglePathMerkleProof (decommit.rs:10-19), SinglePairMerkleProof (folding.rs:20-29) — assembled from
the buffers of rsv_hints_out (include/rsv.h)."""
from __future__ import annotations

import numpy as np

PLONK_COLS = (10, 12, 8)
POSEIDON_COLS = (40, 48, 8)


class PathProof:
    def __init__(self, query, sibling_hashes, columns, depth):
        self.query, self.sibling_hashes, self.columns, self.depth = query, sibling_hashes, columns, depth


class PairProof:
    def __init__(self, query, sibling_hashes, self_columns, siblings_columns, depth):
        self.query, self.sibling_hashes, self.depth = query, sibling_hashes, depth
        self.self_columns, self.siblings_columns = self_columns, siblings_columns


class ProofData:
    def tree_levels(self, t):
        """[(column log size, number of columns)] of commitment tree t, leaf level first (the packing of d_trace_cols)."""
        if t == 3:
            return [(self.M, 8)]
        if self.A == self.B:
            return [(self.A, PLONK_COLS[t] + POSEIDON_COLS[t])]
        return sorted([(self.A, PLONK_COLS[t]), (self.B, POSEIDON_COLS[t])], reverse=True)

    def fill_hints(self, trace_sib, trace_pos, trace_cols, fri_sib, fri_cols):
        """trace_sib [4][nq][M][8], trace_pos [4][nq], trace_cols [4][nq][64], fri_sib [1 + n_inner][nq][M][8],
        fri_cols [1 + n_inner][nq][3][8]; queries in transcript (draw) order."""
        nq, M, A, B = self.nq, self.M, self.A, self.B
        self.raw_trace_cols, self.raw_fri_cols = trace_cols, fri_cols
        self.decommit = []
        for t in range(4):
            dep = M if t == 3 else max(A, B)
            proofs = []
            for i in range(nq):
                columns, off = {}, 0
                for log_size, n in self.tree_levels(t):
                    columns[log_size] = [int(x) for x in trace_cols[t, i, off:off + n]]
                    off += n
                proofs.append(PathProof(int(trace_pos[t, i]), [tuple(int(x) for x in trace_sib[t, i, k]) for k in range(dep)], columns, dep))
            self.decommit.append(proofs)
        data_levels = sorted({M, A, B}, reverse=True)
        self.queries_M = [int(x) for x in trace_pos[3]]
        self.first_layer = []
        for i in range(nq):
            selfc = {ls: tuple(int(x) for x in fri_cols[0, i, c, 0:4]) for c, ls in enumerate(data_levels)}
            sibc = {ls: tuple(int(x) for x in fri_cols[0, i, c, 4:8]) for c, ls in enumerate(data_levels)}
            self.first_layer.append(PairProof(self.queries_M[i], [tuple(int(x) for x in fri_sib[0, i, k]) for k in range(M - 1)], selfc, sibc, M))
        self.inner_layers = {}
        for l in range(self.n_inner):
            ls = M - 1 - l
            proofs = []
            for i in range(nq):
                selfc = {ls: tuple(int(x) for x in fri_cols[1 + l, i, 0, 0:4])}
                sibc = {ls: tuple(int(x) for x in fri_cols[1 + l, i, 0, 4:8])}
                proofs.append(PairProof(self.queries_M[i] >> (M - ls), [tuple(int(x) for x in fri_sib[1 + l, i, k]) for k in range(ls - 1)],
                                        selfc, sibc, ls))
            self.inner_layers[ls] = proofs


def parse_proof(proof: bytes) -> ProofData:
    """bincode layout of PlonkWithPoseidonProof<Poseidon31MerkleHasher>: SURVEY App. A (the parts the circuit allocates)."""
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
    # where a hint sits in the hint buffers (program.py): the column values of tree t at a log size start at
    # trace_col_base inside the query's 64-word row; a pair tree's values at a log size are its fri_col_level-th pair
    d.trace_col_base, d.fri_col_level = {}, {}
    for t in range(4):
        off = 0
        for log_size, n in d.tree_levels(t):
            d.trace_col_base[(t, log_size)] = off
            off += n
    for c, ls in enumerate(sorted({d.M, d.A, d.B}, reverse=True)):
        d.fri_col_level[(0, ls)] = c
    for l in range(n_inner):
        d.fri_col_level[(1 + l, d.M - 1 - l)] = 0
    return d
