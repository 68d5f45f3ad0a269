"""ctypes binding of the CPU oracle (oracle/librsv_oracle.so) — test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RSV_ORACLE_LIB") or os.path.join(ROOT, "oracle", "librsv_oracle.so")  # override: sanitizer build

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


class PcsConfig(ctypes.Structure):
    _fields_ = [("pow_bits", ctypes.c_uint32), ("log_blowup_factor", ctypes.c_uint32),
                ("log_last_layer_degree_bound", ctypes.c_uint32), ("n_queries", ctypes.c_uint32)]


class CfgSet(ctypes.Structure):  # rsv_cfg_set
    _fields_ = [("cfgs", ctypes.POINTER(PcsConfig)), ("n_cfgs", ctypes.c_uint32), ("cfg_of", ctypes.c_void_p)]


class PublicInput(ctypes.Structure):
    _fields_ = [("idx", ctypes.c_uint32), ("value", ctypes.c_uint32 * 4)]


def build():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("rsv_oracle.c", "rsv_emulated.c", "rsv_oracle.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


build()
lib = ctypes.CDLL(LIB)
lib.rsvo_perm_count.restype = ctypes.c_uint64
sz = ctypes.c_size_t
lib.rsvo_poseidon2_permute.argtypes = [_u32p, _u32p, sz]
lib.rsvo_poseidon2_half_permute.argtypes = [_u32p, _u32p, _u8p, _u32p, _u32p, sz]
lib.rsvo_merkle_hash_node.argtypes = [_u32p, _u32p, _u32p, sz, _u32p, sz]
lib.rsvo_merkle_path_root.argtypes = [_u32p, _u32p, _u32p, _u32p, ctypes.c_uint32, _u32p, sz]
lib.rsvo_transcript.argtypes = [_u8p, sz, _u32p, sz]
lib.rsvo_verify_batch.argtypes = [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, _u8p, _u8p]
lib.rsvo_query_values.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz]
lib.rsvo_qm31_mul.argtypes = [_u32p, _u32p, _u32p]
lib.rsvo_qm31_inv.argtypes = [_u32p, _u32p]
lib.rsvo_domain_point.argtypes = [ctypes.c_uint32, ctypes.c_uint32, _u32p]

lib.rsvo_trace_paths.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, _u32p, _u32p, _u32p]

lib.rsvo_fri_paths.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, _u32p, _u32p, _u32p]

STANDARD_INPUTS = [(1, (1, 0, 0, 0)), (2, (0, 1, 0, 0)), (3, (0, 0, 1, 0))]


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def make_inputs(inputs):
    items = list(inputs)
    arr = (PublicInput * max(len(items), 1))()
    for k, (idx, val) in enumerate(items):
        arr[k].idx = idx
        for t in range(4):
            arr[k].value[t] = int(val[t])
    return arr


def poseidon2_permute(states):
    s = _u32(states).reshape(-1, 16)
    out = np.empty_like(s)
    rc = lib.rsvo_poseidon2_permute(s.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), s.shape[0])
    assert rc == 0, rc
    return out


def half_permute(left, right, swap=None):
    l = _u32(left).reshape(-1, 8)
    r = _u32(right).reshape(-1, 8)
    n = l.shape[0]
    sw = None if swap is None else np.ascontiguousarray(swap, dtype=np.uint8)
    rate = np.empty((n, 8), np.uint32)
    cap = np.empty((n, 8), np.uint32)
    rc = lib.rsvo_poseidon2_half_permute(l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p),
                                         None if sw is None else sw.ctypes.data_as(_u8p),
                                         rate.ctypes.data_as(_u32p), cap.ctypes.data_as(_u32p), n)
    assert rc == 0, rc
    return rate, cap


def hash_node(children, cols):
    c = _u32(cols)
    if c.ndim == 1:
        c = c.reshape(1, -1)
    n, n_cols = c.shape
    out = np.empty((n, 8), np.uint32)
    if children is None:
        lp = rp = None
    else:
        l = _u32(children[0]).reshape(n, 8)
        r = _u32(children[1]).reshape(n, 8)
        lp, rp = l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p)
    rc = lib.rsvo_merkle_hash_node(lp, rp, c.ctypes.data_as(_u32p) if n_cols else None, n_cols,
                                   out.ctypes.data_as(_u32p), n)
    assert rc == 0, rc
    return out


def merkle_path_root(query, siblings, cols, n_cols_at):
    q = _u32(query).reshape(-1)
    n = q.shape[0]
    depth = len(n_cols_at) - 1
    sib = _u32(siblings).reshape(n, depth, 8)
    nca = _u32(n_cols_at)
    c = _u32(cols).reshape(n, int(nca.sum()))
    out = np.empty((n, 8), np.uint32)
    rc = lib.rsvo_merkle_path_root(q.ctypes.data_as(_u32p), sib.ctypes.data_as(_u32p), c.ctypes.data_as(_u32p),
                                   nca.ctypes.data_as(_u32p), depth, out.ctypes.data_as(_u32p), n)
    assert rc == 0, rc
    return out


def transcript_raw(proof: bytes):
    b = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(1024, np.uint32)
    rc = lib.rsvo_transcript(b.ctypes.data_as(_u8p), len(proof), out.ctypes.data_as(_u32p), out.size)
    assert rc == 0, rc
    return out


def pack(proofs):
    offsets = np.zeros(len(proofs) + 1, np.uint64)
    if proofs:
        offsets[1:] = np.cumsum([len(p) for p in proofs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(proofs), dtype=np.uint8) if proofs else np.zeros(0, np.uint8)
    return blob, offsets


def make_cfg_set(cfg, n):
    """cfg: one PcsConfig-like object or a sequence of n of them -> (CfgSet, keep-alive tuple)."""
    if cfg is None:
        raise TypeError("a PcsConfig (or one per proof) is required")
    per = [cfg] * 1 if hasattr(cfg, "pow_bits") else list(cfg)
    key = lambda c: (int(c.pow_bits), int(c.log_blowup_factor), int(c.log_last_layer_degree_bound), int(c.n_queries))  # noqa: E731
    table, index = [], {}
    cfg_of = np.zeros(max(n, 1), np.uint8)
    if not hasattr(cfg, "pow_bits"):
        assert len(per) == n
    for i, c in enumerate(per):
        if key(c) not in index:
            index[key(c)] = len(table)
            table.append(key(c))
        cfg_of[i] = index[key(c)]
    if not table:
        table = [(0, 0, 0, 0)]
    arr = (PcsConfig * len(table))(*[PcsConfig(*k) for k in table])
    of = cfg_of if len(table) > 1 else None
    cs = CfgSet(ctypes.cast(arr, ctypes.POINTER(PcsConfig)), len(table), of.ctypes.data if of is not None else None)
    return cs, (arr, of)


def verify_batch(proofs, cfg, inputs=STANDARD_INPUTS):
    blob, offsets = pack(proofs)
    n = len(proofs)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    pi = make_inputs(inputs)
    cs, _keep = make_cfg_set(cfg, n)
    rc = lib.rsvo_verify_batch(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n,
                               ctypes.byref(cs), pi, len(list(inputs)),
                               accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p))
    if rc != 0:
        raise RuntimeError(f"rsvo_verify_batch -> {rc}")
    return accept, reason


def header_cfg(proof: bytes) -> PcsConfig:
    """The configuration words serialized in a proof (words 10..13).  For tests that build batches from
    fixtures whose header is known to equal the manifest's configuration — never a substitute for the caller's
    configuration in the product."""
    w = np.frombuffer(proof[:56], dtype=np.uint32)
    return PcsConfig(int(w[10]), int(w[11]), int(w[12]), int(w[13]))


def perm_count(proof: bytes, inputs=STANDARD_INPUTS):
    lib.rsvo_perm_count_reset()
    verify_batch([proof], header_cfg(proof), inputs)
    return int(lib.rsvo_perm_count())


def qm31_mul(a, b):
    a, b = _u32(a), _u32(b)
    out = np.zeros(4, np.uint32)
    lib.rsvo_qm31_mul(a.ctypes.data_as(_u32p), b.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p))
    return out


def qm31_inv(a):
    a = _u32(a)
    out = np.zeros(4, np.uint32)
    lib.rsvo_qm31_inv(a.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p))
    return out


def domain_point(log_size, q):
    out = np.zeros(2, np.uint32)
    lib.rsvo_domain_point(log_size, q, out.ctypes.data_as(_u32p))
    return int(out[0]), int(out[1])


def field_op(op, a, b=None):
    x = _u32(a).reshape(-1, 4)
    y = None if b is None else _u32(b).reshape(-1, 4)
    out = np.empty_like(x)
    lib.rsvo_field_op.restype = ctypes.c_int
    lib.rsvo_field_op.argtypes = [ctypes.c_int, _u32p, _u32p, _u32p, ctypes.c_size_t]
    rc = lib.rsvo_field_op(op, x.ctypes.data_as(_u32p), None if y is None else y.ctypes.data_as(_u32p),
                           out.ctypes.data_as(_u32p), x.shape[0])
    assert rc == 0, rc
    return out


def domain_points(log_size, q):
    return np.array([domain_point(log_size, int(v) & ((1 << log_size) - 1)) for v in np.asarray(q).reshape(-1)], np.uint32)


def line_eval(coeffs, x):
    c = _u32(coeffs).reshape(-1, 4)
    log_n = int(c.shape[0]).bit_length() - 1
    xx = _u32(x).reshape(-1)
    out = np.empty((xx.size, 4), np.uint32)
    lib.rsvo_line_eval.restype = ctypes.c_int
    lib.rsvo_line_eval.argtypes = [_u32p, ctypes.c_uint32, _u32p, _u32p, ctypes.c_size_t]
    rc = lib.rsvo_line_eval(c.ctypes.data_as(_u32p), log_n, xx.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), xx.size)
    assert rc == 0, rc
    return out


def grind_nonce(proof: bytes, want_duplicate_query: bool = False, start: int = 0, max_tries: int = 1 << 34) -> bytes:
    """Re-grind the proof-of-work nonce of a (modified) proof so that it passes the PoW check again — optionally with a
    duplicate query position — and return the patched proof.  Test-vector helper (rsvo_grind_nonce)."""
    b = np.frombuffer(proof, dtype=np.uint8)
    nonce = ctypes.c_uint64(0)
    lib.rsvo_grind_nonce.restype = ctypes.c_int
    lib.rsvo_grind_nonce.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    rc = lib.rsvo_grind_nonce(b.ctypes.data_as(_u8p), len(proof), start, max_tries, 1 if want_duplicate_query else 0, ctypes.byref(nonce))
    if rc != 0:
        raise RuntimeError(f"rsvo_grind_nonce -> {rc}")
    pos = 4 * proof_layout(proof)["nonce_word"]
    out = bytearray(proof)
    out[pos:pos + 8] = int(nonce.value).to_bytes(8, "little")
    return bytes(out)


def splitmix64(seed, i):
    """splitmix64 stream used for seeded tampering."""
    mask = (1 << 64) - 1
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & mask
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    return z ^ (z >> 31)


def tamper(proof: bytes, i: int, seed: int = 0xC0FFEE) -> bytes:
    """Flip the low bit of one byte at offset 60 + splitmix64(seed, i) % (len - 68)."""
    b = bytearray(proof)
    off = 60 + splitmix64(seed, i) % (len(b) - 68)
    b[off] ^= 1
    return bytes(b)


def trace_paths(proof: bytes, n_queries: int, max_log: int, inputs=STANDARD_INPUTS):
    """-> (sib uint32[4, nq, M, 8], pos uint32[4, nq], depth uint32[4])."""
    b = np.frombuffer(proof, dtype=np.uint8)
    sib = np.zeros((4, n_queries, max_log, 8), np.uint32)
    pos = np.zeros((4, n_queries), np.uint32)
    depth = np.zeros(4, np.uint32)
    nq = np.zeros(1, np.uint32)
    pi = make_inputs(inputs)
    rc = lib.rsvo_trace_paths(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), sib.ctypes.data_as(_u32p),
                              sib.size, pos.ctypes.data_as(_u32p), depth.ctypes.data_as(_u32p), nq.ctypes.data_as(_u32p))
    if rc != 0:
        raise RuntimeError(f"rsvo_trace_paths -> {rc}")
    assert int(nq[0]) == n_queries
    return sib, pos, depth


def poseidon_flow(proof: bytes, inputs=STANDARD_INPUTS):
    """PoseidonFlow of the circuit that verifies `proof` (rsvo_poseidon_flow): uint32[count, 33] =
    left8 | right8 | out_rate8 | out_cap8 | swap per Poseidon2HalfVar::permute invocation, in invocation order."""
    b = np.frombuffer(proof, dtype=np.uint8)
    pi = make_inputs(inputs)
    lib.rsvo_poseidon_flow.restype = ctypes.c_int
    lib.rsvo_poseidon_flow.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, ctypes.POINTER(sz)]
    count = sz(0)
    rc = lib.rsvo_poseidon_flow(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), None, 0, ctypes.byref(count))
    if rc not in (0, -4):
        raise RuntimeError(f"rsvo_poseidon_flow -> {rc}")
    out = np.zeros((count.value, 33), np.uint32)
    rc = lib.rsvo_poseidon_flow(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), out.ctypes.data_as(_u32p), count.value,
                                ctypes.byref(count))
    if rc != 0:
        raise RuntimeError(f"rsvo_poseidon_flow -> {rc}")
    return out


ROWS_PER_INVOCATION = 6


def flow_log_size(n_invocations: int) -> int:
    """log2 of the Poseidon trace a flow of n invocations becomes: padded to a multiple of 16, at least 2 * N_LANES = 32
    (PlonkWithPoseidonConstraintSystem::pad, constraint_system/src/plonk_with_poseidon.rs:282-300), SIX trace rows per
    invocation, next power of two.  Six rows = the shape of the Poseidon AIR the verifier evaluates
    (components/recursive/composition/src/poseidon.rs:73-241): one `is_first` row (the initial external matrix), two
    `is_full` rows of two full rounds each, one partial row (all 14 partial rounds), two more full rows.  The count is
    also forced by the fixtures: 6 is the only integer for which every one of the 15 consecutive fixture pairs of the
    reference (tests/test_oracle.py::test_poseidon_flow_count_predicts_next_level) lands on the header's log size —
    level2-1's flow needs <= 6.23 rows per invocation, level7-1's > 5.82.  (SURVEY §0 guessed 8 from small_proof.bin
    alone, whose 32-invocation minimum fits 2^8 rows for 5..8.)"""
    padded = max(32, -(-n_invocations // 16) * 16)
    return (ROWS_PER_INVOCATION * padded - 1).bit_length()


def trace_cols(proof: bytes, inputs=STANDARD_INPUTS):
    """-> uint32[4, nq, 64]: SinglePathMerkleProof::columns per query, leaf-level columns then lower-level ones."""
    b = np.frombuffer(proof, dtype=np.uint8)
    cols = np.zeros(4 * 128 * 64, np.uint32)
    nq = np.zeros(1, np.uint32)
    pi = make_inputs(inputs)
    lib.rsvo_trace_cols.restype = ctypes.c_int
    lib.rsvo_trace_cols.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, _u32p]
    rc = lib.rsvo_trace_cols(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), cols.ctypes.data_as(_u32p), cols.size,
                             nq.ctypes.data_as(_u32p))
    if rc != 0:
        raise RuntimeError(f"rsvo_trace_cols -> {rc}")
    n = int(nq[0])
    return cols[:4 * n * 64].reshape(4, n, 64).copy()


def fri_folded(proof: bytes, inputs=STANDARD_INPUTS):
    """-> uint32[3, nq, 4]: FirstLayerHints::folded_evals_by_column per column log size (descending) and query."""
    b = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(3 * 128 * 4, np.uint32)
    ns, nq = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    pi = make_inputs(inputs)
    lib.rsvo_fri_folded.restype = ctypes.c_int
    lib.rsvo_fri_folded.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, _u32p, _u32p]
    rc = lib.rsvo_fri_folded(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), out.ctypes.data_as(_u32p), out.size,
                             ns.ctypes.data_as(_u32p), nq.ctypes.data_as(_u32p))
    if rc != 0:
        raise RuntimeError(f"rsvo_fri_folded -> {rc}")
    n = int(nq[0])
    return out[:3 * n * 4].reshape(3, n, 4).copy()


def fri_paths(proof: bytes, n_queries: int, max_log: int, n_trees: int, inputs=STANDARD_INPUTS):
    """-> (sib uint32[n_trees, nq, M, 8], cols uint32[n_trees, nq, 3, 8])."""
    b = np.frombuffer(proof, dtype=np.uint8)
    sib = np.zeros((n_trees, n_queries, max_log, 8), np.uint32)
    cols = np.zeros((n_trees, n_queries, 3, 8), np.uint32)
    nt = np.zeros(1, np.uint32)
    nq = np.zeros(1, np.uint32)
    pi = make_inputs(inputs)
    rc = lib.rsvo_fri_paths(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), sib.ctypes.data_as(_u32p), sib.size,
                            cols.ctypes.data_as(_u32p), nt.ctypes.data_as(_u32p), nq.ctypes.data_as(_u32p))
    if rc != 0:
        raise RuntimeError(f"rsvo_fri_paths -> {rc}")
    assert int(nt[0]) == n_trees and int(nq[0]) == n_queries
    return sib, cols


def oods_eval(samples, params):
    sm = _u32(samples).reshape(-1, 142, 4)
    pr = _u32(params).reshape(-1, 26)
    out = np.empty((sm.shape[0], 8), np.uint32)
    lib.rsvo_oods_eval.restype = ctypes.c_int
    lib.rsvo_oods_eval.argtypes = [_u32p, _u32p, _u32p, sz]
    rc = lib.rsvo_oods_eval(sm.ctypes.data_as(_u32p), pr.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), sm.shape[0])
    assert rc == 0, rc
    return out


def query_dump(proof: bytes, inputs=STANDARD_INPUTS):
    """-> uint32[nq, 4 * (8 + n_inner)]: the layout of rsv_hints_out::d_query_values for one proof."""
    b = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(128 * 4 * (8 + 29), np.uint32)
    ni, nq = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    pi = make_inputs(inputs)
    lib.rsvo_query_dump.restype = ctypes.c_int
    lib.rsvo_query_dump.argtypes = [_u8p, sz, ctypes.POINTER(PublicInput), sz, _u32p, sz, _u32p, _u32p]
    rc = lib.rsvo_query_dump(b.ctypes.data_as(_u8p), len(proof), pi, len(list(inputs)), out.ctypes.data_as(_u32p), out.size,
                             ni.ctypes.data_as(_u32p), nq.ctypes.data_as(_u32p))
    if rc != 0:
        raise RuntimeError(f"rsvo_query_dump -> {rc}")
    stride = 4 * (8 + int(ni[0]))
    return out[:int(nq[0]) * stride].reshape(int(nq[0]), stride).copy()


def sampled_values(proof: bytes) -> np.ndarray:
    """The 142 sampled values of a proof, flattened tree-major / column-major / sample-minor: uint32[142, 4]."""
    w = np.frombuffer(proof, dtype=np.uint32)
    pos, out = 49 + 2, []
    for t, ncols in enumerate((50, 60, 16, 8)):
        assert int(w[pos]) == ncols
        pos += 2
        for c in range(ncols):
            ns = int(w[pos]); pos += 2
            for _ in range(ns):
                out.append(w[pos:pos + 4].copy()); pos += 4
    assert pos == 895 and len(out) == 142
    return np.stack(out)


def proof_layout(proof: bytes):
    """Word offsets of the variable part of a proof: FRI layer commitments and the position of
    every u64 length prefix (`prefixes`: (word offset, count, what))."""
    w = np.frombuffer(proof, dtype=np.uint32)
    prefixes = []
    pos = 895
    prefixes.append((pos, int(w[pos]), "decommitments")); pos += 2
    for t in range(4):
        nh = int(w[pos]); prefixes.append((pos, nh, f"hash_witness[{t}]")); pos += 2 + 8 * nh
        prefixes.append((pos, int(w[pos]), f"column_witness[{t}]")); pos += 2
    prefixes.append((pos, int(w[pos]), "queried_values")); pos += 2
    for t in range(4):
        nv = int(w[pos]); prefixes.append((pos, nv, f"queried_values[{t}]")); pos += 2 + nv
    nonce_word = pos
    pos += 2  # proof-of-work nonce
    layers = []

    def layer(pos, name):
        nw = int(w[pos]); prefixes.append((pos, nw, name + ".fri_witness")); pos += 2 + 4 * nw
        nh = int(w[pos]); prefixes.append((pos, nh, name + ".hash_witness")); pos += 2 + 8 * nh
        prefixes.append((pos, int(w[pos]), name + ".column_witness")); pos += 2
        return pos + 8, w[pos:pos + 8].copy()

    pos, c0 = layer(pos, "first")
    layers.append(c0)
    n_inner = int(w[pos]); prefixes.append((pos, n_inner, "inner_layers")); pos += 2
    for i in range(n_inner):
        pos, ci = layer(pos, f"inner[{i}]")
        layers.append(ci)
    prefixes.append((pos, int(w[pos]), "last_layer_poly"))
    return {"lp": int(w[0]), "lq": int(w[1]), "blowup": int(w[11]), "log_last": int(w[12]), "nq": int(w[13]),
            "fri_commitments": layers, "n_inner": n_inner, "prefixes": prefixes, "nonce_word": nonce_word}


def split_variable_part(proof: bytes):
    """Parse the variable part of a proof (from word 895) into Python lists of uint32 arrays so that a
    test can re-serialize a structurally valid but inconsistent proof (see join_variable_part)."""
    w = np.frombuffer(proof, dtype=np.uint32)
    pos = 895 + 2
    d = {"head": w[:895].copy(), "hash_witness": [], "queried_values": [], "layers": []}
    for _ in range(4):
        nh = int(w[pos]); pos += 2
        d["hash_witness"].append([w[pos + 8 * k:pos + 8 * k + 8].copy() for k in range(nh)]); pos += 8 * nh + 2
    pos += 2
    for _ in range(4):
        nv = int(w[pos]); pos += 2
        d["queried_values"].append(list(w[pos:pos + nv])); pos += nv
    d["nonce"] = w[pos:pos + 2].copy(); pos += 2

    def layer(pos):
        nw = int(w[pos]); pos += 2
        wit = [w[pos + 4 * k:pos + 4 * k + 4].copy() for k in range(nw)]; pos += 4 * nw
        nh = int(w[pos]); pos += 2
        hw = [w[pos + 8 * k:pos + 8 * k + 8].copy() for k in range(nh)]; pos += 8 * nh + 2
        return pos + 8, {"fri_witness": wit, "hash_witness": hw, "commitment": w[pos:pos + 8].copy()}

    pos, first = layer(pos)
    d["layers"].append(first)
    n_inner = int(w[pos]); pos += 2
    for _ in range(n_inner):
        pos, l = layer(pos)
        d["layers"].append(l)
    nl = int(w[pos]); pos += 2
    d["last"] = [w[pos + 4 * k:pos + 4 * k + 4].copy() for k in range(nl)]; pos += 4 * nl
    d["tail"] = w[pos:].copy()
    return d


def join_variable_part(d) -> bytes:
    def u64(n):
        return np.array([n & 0xFFFFFFFF, n >> 32], dtype=np.uint32)

    def cat(items, width):
        return np.concatenate([np.asarray(x, dtype=np.uint32).reshape(width) for x in items]) if items else np.zeros(0, np.uint32)

    out = [d["head"], u64(4)]
    for t in range(4):
        out += [u64(len(d["hash_witness"][t])), cat(d["hash_witness"][t], 8), u64(0)]
    out.append(u64(4))
    for t in range(4):
        out += [u64(len(d["queried_values"][t])), np.asarray(d["queried_values"][t], dtype=np.uint32)]
    out.append(d["nonce"])
    for i, l in enumerate(d["layers"]):
        if i == 1:
            out.append(u64(len(d["layers"]) - 1))
        out += [u64(len(l["fri_witness"])), cat(l["fri_witness"], 4), u64(len(l["hash_witness"])), cat(l["hash_witness"], 8),
                u64(0), l["commitment"]]
    if len(d["layers"]) == 1:
        out.append(u64(0))
    out += [u64(len(d["last"])), cat(d["last"], 4), d["tail"]]
    return np.concatenate([np.asarray(x, dtype=np.uint32) for x in out]).tobytes()


class _Cheap:
    """copy.deepcopy of a parsed proof costs ~0.2 s for the 80-query shapes (tens of thousands of small arrays) and the
    mutant generators make ~150 copies per fixture: the lists are copied, the arrays (never modified in place) shared."""

    @staticmethod
    def deepcopy(d):
        return {"head": d["head"], "hash_witness": [list(x) for x in d["hash_witness"]],
                "queried_values": [list(x) for x in d["queried_values"]], "nonce": d["nonce"],
                "layers": [{"fri_witness": list(l["fri_witness"]), "hash_witness": list(l["hash_witness"]), "commitment": l["commitment"]}
                           for l in d["layers"]],
                "last": list(d["last"]), "tail": d["tail"]}


def structural_mutants(proof: bytes):
    """Structurally valid re-serializations with one list one element too short / too long, or with elements moved:
    they parse, and must fail in the stage that consumes the list.  Returns [(tag, bytes)]."""
    copy = _Cheap
    base = split_variable_part(proof)
    out = []

    def emit(tag, d):
        out.append((tag, join_variable_part(d)))

    zero8, zero4 = np.zeros(8, np.uint32), np.zeros(4, np.uint32)
    for t in range(4):
        d = copy.deepcopy(base); d["hash_witness"][t].pop(); emit(f"hw[{t}]-1", d)
        d = copy.deepcopy(base); d["hash_witness"][t].append(zero8); emit(f"hw[{t}]+1", d)
        d = copy.deepcopy(base); d["hash_witness"][t].insert(0, d["hash_witness"][t].pop()); emit(f"hw[{t}] rotated", d)
        d = copy.deepcopy(base); d["queried_values"][t].pop(); emit(f"qv[{t}]-1", d)
        d = copy.deepcopy(base); d["queried_values"][t].append(np.uint32(7)); emit(f"qv[{t}]+1", d)
        d = copy.deepcopy(base); d["queried_values"][t] = d["queried_values"][t][: len(d["queried_values"][t]) // 2]; emit(f"qv[{t}] halved", d)
        d = copy.deepcopy(base); d["hash_witness"][t] = []; emit(f"hw[{t}] empty", d)
    d = copy.deepcopy(base); d["hash_witness"][1].append(d["hash_witness"][0].pop()); emit("hw[0]->hw[1]", d)
    d = copy.deepcopy(base); d["queried_values"][1].append(d["queried_values"][0].pop()); emit("qv[0]->qv[1]", d)
    for i in range(len(base["layers"])):
        for key, z in (("fri_witness", zero4), ("hash_witness", zero8)):
            if base["layers"][i][key]:
                d = copy.deepcopy(base); d["layers"][i][key].pop(); emit(f"layer[{i}].{key}-1", d)
                d = copy.deepcopy(base); d["layers"][i][key] = []; emit(f"layer[{i}].{key} empty", d)
            d = copy.deepcopy(base); d["layers"][i][key].append(z); emit(f"layer[{i}].{key}+1", d)
    if len(base["layers"]) > 2:
        d = copy.deepcopy(base); d["layers"].pop(); emit("inner layers -1", d)
        d = copy.deepcopy(base); d["layers"].append(dict(d["layers"][-1])); emit("inner layers +1", d)
        d = copy.deepcopy(base); d["layers"][1], d["layers"][2] = d["layers"][2], d["layers"][1]; emit("inner layers swapped", d)
    d = copy.deepcopy(base); d["last"].pop(); emit("last-1", d)
    d = copy.deepcopy(base); d["last"].append(zero4); emit("last+1", d)
    d = copy.deepcopy(base); d["last"] = d["last"] + d["last"]; emit("last doubled", d)
    d = copy.deepcopy(base); d["last"] = []; emit("last empty", d)
    return out


def noncanonical_structural_mutants(proof: bytes):
    """Proofs that are wrong twice: a witness list of the wrong length (the stage that consumes it rejects the proof and
    never reads all of it) AND a non-canonical word in the part nobody consumes, or in an element that moved.  The
    verdict is reject either way; the REASON must be PARSE (a non-canonical field element outranks every later
    stage), which the product can only say if its canonicity check reaches words no stage reads (csrc/layout.hpp:
    F_RESCAN).  Returns [(tag, bytes)]."""
    copy = _Cheap
    base = split_variable_part(proof)
    out = []
    P_, MAXW = 0x7FFFFFFF, 0xFFFFFFFF

    def emit(tag, d):
        out.append((tag, join_variable_part(d)))

    def bad8(k, val):
        h = np.zeros(8, np.uint32); h[k] = val
        return h

    for t in range(4):
        d = copy.deepcopy(base); d["hash_witness"][t].append(bad8(7, P_)); emit(f"hw[{t}]+P", d)
        d = copy.deepcopy(base); d["hash_witness"][t] += [bad8(0, 5), bad8(3, MAXW), bad8(1, 6)]; emit(f"hw[{t}]+3 (middle one bad)", d)
        if len(base["hash_witness"][t]) > 3:
            d = copy.deepcopy(base); d["hash_witness"][t].pop(0); d["hash_witness"][t][-1] = bad8(2, P_); emit(f"hw[{t}]-1, last bad", d)
        d = copy.deepcopy(base); d["queried_values"][t].append(np.uint32(P_)); emit(f"qv[{t}]+P", d)
        qv = list(base["queried_values"][t])
        d = copy.deepcopy(base); d["queried_values"][t] = qv + qv + qv; d["queried_values"][t][len(qv) + len(qv) // 2] = np.uint32(MAXW)
        emit(f"qv[{t}] tripled, middle bad", d)
        d = copy.deepcopy(base); d["queried_values"][t] = qv[: len(qv) // 2 + 3]; d["queried_values"][t][-2] = np.uint32(P_)
        emit(f"qv[{t}] halved ragged, bad near the cut", d)
    for i in range(len(base["layers"])):
        d = copy.deepcopy(base); d["layers"][i]["fri_witness"].append(np.array([1, 2, P_, 4], np.uint32)); emit(f"layer[{i}].fri_witness+P", d)
        d = copy.deepcopy(base); d["layers"][i]["hash_witness"].append(bad8(4, MAXW)); emit(f"layer[{i}].hash_witness+bad", d)
        if len(base["layers"][i]["hash_witness"]) > 2:
            d = copy.deepcopy(base); d["layers"][i]["hash_witness"].pop(0); d["layers"][i]["hash_witness"][-1] = bad8(6, P_)
            emit(f"layer[{i}].hash_witness-1, last bad", d)
        if len(base["layers"][i]["fri_witness"]) > 1:
            d = copy.deepcopy(base); d["layers"][i]["fri_witness"].pop(0); d["layers"][i]["fri_witness"][-1] = np.array([P_, 0, 0, 0], np.uint32)
            emit(f"layer[{i}].fri_witness-1, last bad", d)
    return out


# ---------------------------------------------------------------------------- emulated Poseidon2 (oracle/rsv_emulated.c)
VAR_FIXED, VAR_WITNESS, VAR_CONSTANT, VAR_GATE = 0, 1, 2, 3
lib.rsvo_ecs_new.restype = ctypes.c_void_p
lib.rsvo_ecs_free.argtypes = [ctypes.c_void_p]
lib.rsvo_ecs_n_vars.argtypes = [ctypes.c_void_p]
lib.rsvo_ecs_n_vars.restype = sz
lib.rsvo_ecs_n_rows.argtypes = [ctypes.c_void_p]
lib.rsvo_ecs_n_rows.restype = sz
lib.rsvo_ecs_new_witness_m31.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
lib.rsvo_ecs_new_witness_m31.restype = ctypes.c_uint32
lib.rsvo_ecs_new_witness_qm31.argtypes = [ctypes.c_void_p, _u32p]
lib.rsvo_ecs_new_witness_qm31.restype = ctypes.c_uint32
lib.rsvo_ecs_qm31_from_m31.argtypes = [ctypes.c_void_p, _u32p]
lib.rsvo_ecs_qm31_from_m31.restype = ctypes.c_uint32
lib.rsvo_ecs_permute_emulated.argtypes = [ctypes.c_void_p, _u32p, _u32p, ctypes.c_int, ctypes.c_uint32, _u32p]
lib.rsvo_ecs_check_arithmetics.argtypes = [ctypes.c_void_p]
lib.rsvo_ecs_check_arithmetics.restype = sz
lib.rsvo_ecs_export.argtypes = [ctypes.c_void_p, _u32p, _u8p, _u32p]
lib.rsvo_ecs_set_vars.argtypes = [ctypes.c_void_p, sz, _u32p, sz]


class EmulatedCS:
    """The oracle's minimal Plonk-without-Poseidon constraint system with the emulated Poseidon2 gadget on it."""

    def __init__(self):
        self.h = ctypes.c_void_p(lib.rsvo_ecs_new())

    def __del__(self):
        if getattr(self, "h", None):
            lib.rsvo_ecs_free(self.h)
            self.h = None

    @property
    def n_vars(self):
        return lib.rsvo_ecs_n_vars(self.h)

    @property
    def n_rows(self):
        return lib.rsvo_ecs_n_rows(self.h)

    def witness_m31(self, v):
        return lib.rsvo_ecs_new_witness_m31(self.h, int(v))

    def witness_qm31(self, v4):
        a = _u32(v4)
        return lib.rsvo_ecs_new_witness_qm31(self.h, a.ctypes.data_as(_u32p))

    def qm31_from_m31(self, vars4):
        a = _u32(vars4)
        return lib.rsvo_ecs_qm31_from_m31(self.h, a.ctypes.data_as(_u32p))

    def permute(self, left2, right2, swap_bit_var=None):
        """swap_bit_var None = is_swap None; else the variable index of the bit (is_swap = Some((value, var)))."""
        l, r, out = _u32(left2), _u32(right2), np.zeros(4, np.uint32)
        rc = lib.rsvo_ecs_permute_emulated(self.h, l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p),
                                           0 if swap_bit_var is None else 1, 0 if swap_bit_var is None else swap_bit_var,
                                           out.ctypes.data_as(_u32p))
        if rc != 0:
            raise RuntimeError(f"rsvo_ecs_permute_emulated: {rc}")
        return out

    def check_arithmetics(self) -> int:
        """0 = every row satisfies its gate equation, else 1 + the first failing row."""
        return lib.rsvo_ecs_check_arithmetics(self.h)

    def export(self):
        nv, nr = self.n_vars, self.n_rows
        v, k, r = np.zeros((nv, 4), np.uint32), np.zeros(nv, np.uint8), np.zeros((nr, 7), np.uint32)
        lib.rsvo_ecs_export(self.h, v.ctypes.data_as(_u32p), k.ctypes.data_as(_u8p), r.ctypes.data_as(_u32p))
        return v, k, r

    def set_vars(self, first, vals):
        a = _u32(vals).reshape(-1, 4)
        rc = lib.rsvo_ecs_set_vars(self.h, first, a.ctypes.data_as(_u32p), len(a))
        if rc != 0:
            raise RuntimeError(f"rsvo_ecs_set_vars: {rc}")


def emulated_rows(left8, right8, swap):
    """Steady-state gate values of n emulated permutations, as the product lays them out ([n][416][4]; rows 0..11 are
    the swap rows, zero when swap[p] == 0; rows 413..415 are zero padding).  swap[p]: 0 = None, 1 = Some((false, _)), 2 = Some((true, _)).  One warm-up
    call caches the constants first, as every call but a circuit's first finds them."""
    left8, right8 = _u32(left8).reshape(-1, 8), _u32(right8).reshape(-1, 8)
    n = len(left8)
    cs = EmulatedCS()
    w = [cs.witness_qm31([0, 0, 0, 0]) for _ in range(4)]
    cs.permute(w[:2], w[2:], None)
    cs.permute(w[:2], w[2:], cs.witness_m31(0))
    out = np.zeros((n, 416, 4), np.uint32)
    spans = []
    for p in range(n):
        l = [cs.witness_qm31(left8[p, :4]), cs.witness_qm31(left8[p, 4:])]
        r = [cs.witness_qm31(right8[p, :4]), cs.witness_qm31(right8[p, 4:])]
        bit = None if swap[p] == 0 else cs.witness_m31(int(swap[p]) - 1)
        first = cs.n_vars
        cs.permute(l, r, bit)
        spans.append((first, cs.n_vars, int(swap[p])))
    v, k, _ = cs.export()
    for p, (a, b, sw) in enumerate(spans):
        assert not (k[a:b] == VAR_CONSTANT).any()
        rows = v[a:b]
        assert len(rows) == (413 if sw else 401)
        out[p, (0 if sw else 12):413] = rows
    assert cs.check_arithmetics() == 0
    return out, cs, spans
