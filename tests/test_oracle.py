"""CPU tests of the oracle against the reference's own known answers:
the literal Poseidon2 KAT, the hasher/channel/transcript checkpoints of SURVEY App. C and the
accept/reject behaviour of every proof fixture under the config written in the reference source."""
import sys

import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.conftest import fixture_cfg, load_manifest, read_proof

P = 0x7FFFFFFF

# primitives/poseidon31/src/implementation.rs:157-172
KAT_OUT = [260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943,
           1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264]


def test_poseidon2_kat():
    out = ob.poseidon2_permute(np.arange(16, dtype=np.uint32))
    assert out[0].tolist() == KAT_OUT


def test_poseidon2_rejects_noncanonical():
    s = np.zeros(16, np.uint32)
    s[3] = P
    out = np.zeros(16, np.uint32)
    rc = ob.lib.rsvo_poseidon2_permute(s.ctypes.data_as(ob._u32p), out.ctypes.data_as(ob._u32p), 1)
    assert rc == -5


def test_hash_node_checkpoints():
    # SURVEY App. C
    assert ob.hash_node(None, [1, 2, 3, 4, 5])[0].tolist() == [
        557709851, 1113733662, 222169927, 1376019790, 387901840, 1087892516, 628125718, 969660801]
    l, r = np.arange(1, 9), np.arange(9, 17)
    assert ob.hash_node((l, r), np.zeros((1, 0)))[0].tolist() == [
        164793487, 387042994, 621688597, 428853092, 1214488792, 1623406829, 1918424220, 1537261691]
    assert ob.hash_node((l, r), [7, 7, 7])[0].tolist() == [
        2077916493, 57586551, 1709117860, 800174306, 352135528, 1590574078, 1798659285, 1176940757]


def test_half_permute_swap_semantics():
    # primitives/poseidon31/src/lib.rs:293-309: swap => state = right || left
    rng = np.random.default_rng(1)
    l = rng.integers(0, P, (4, 8), dtype=np.uint32)
    r = rng.integers(0, P, (4, 8), dtype=np.uint32)
    rate, cap = ob.half_permute(l, r, [0, 1, 0, 1])
    full = ob.poseidon2_permute(np.concatenate([l, r], axis=1))
    full_sw = ob.poseidon2_permute(np.concatenate([r, l], axis=1))
    for i, sw in enumerate([0, 1, 0, 1]):
        want = full_sw[i] if sw else full[i]
        assert rate[i].tolist() == want[:8].tolist() and cap[i].tolist() == want[8:].tolist()


def test_half_permute_known_answer_with_swap():
    """The reference's second literal vector (primitives/poseidon31/src/emulated.rs:236-275): permute(0..8 | 8..16)
    without swap, with swap bit 0, and permute(8..16 | 0..8) with swap bit 1 all give the Poseidon2 KAT."""
    kat = np.array([260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943, 1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264], np.uint32)
    lo, hi = np.arange(8, dtype=np.uint32)[None], np.arange(8, 16, dtype=np.uint32)[None]
    for l, r, sw in ((lo, hi, None), (lo, hi, [0]), (hi, lo, [1])):
        rate, cap = ob.half_permute(l, r, sw)
        assert np.concatenate([rate[0], cap[0]]).tolist() == kat.tolist()


def test_sponge_matches_manual_chain():
    # primitives/merkle/src/lib.rs:141-181 for the column lengths the reference tests (7/13/16/17/21/25)
    rng = np.random.default_rng(0)
    for n in (7, 13, 16, 17, 21, 25):
        cols = rng.integers(0, P, n, dtype=np.uint32)
        d = np.zeros(8, np.uint32)
        for off in range(0, n, 8):
            chunk = np.zeros(8, np.uint32)
            k = min(8, n - off)
            chunk[:k] = cols[off:off + k]
            d = ob.poseidon2_permute(np.concatenate([chunk, d]))[0][8:]
        leaf = ob.poseidon2_permute(np.concatenate([np.zeros(8, np.uint32), d]))[0][:8]
        assert ob.hash_node(None, cols)[0].tolist() == leaf.tolist()


def test_small_proof_transcript_checkpoints():
    out = ob.transcript_raw(read_proof("small_proof.bin"))
    assert out[0] == 0 and out[1] == 8 and out[2] == 16 and out[3] == 15
    assert out[4:8].tolist() == [1211683141, 437669427, 409200369, 1127771350]       # z
    assert out[8:12].tolist() == [608237629, 60905622, 1129253272, 1937554417]        # alpha
    assert out[12:16].tolist() == [510535785, 709795745, 2021304333, 468388088]       # random_coeff
    assert out[16:20].tolist() == [432538781, 1881392761, 1838851372, 291147612]      # oods t
    assert out[28:32].tolist() == [258757294, 1276317760, 1536227746, 6873968]        # after sampled values
    assert out[40:44].tolist() == [2118644044, 1562770230, 410003546, 1078681992]     # fri_alpha[0]
    assert out[68:72].tolist() == [1686932136, 243723819, 74374586, 128365204]        # fri_alpha[7]
    q = out[72:88] & ((1 << 15) - 1)
    assert q.tolist() == [3311, 10908, 19594, 28340, 20569, 32639, 20328, 27419, 4097, 16595, 24642, 2474,
                          29786, 20017, 13022, 23721]


@pytest.mark.parametrize("entry", load_manifest(), ids=lambda e: e["file"])
def test_fixture_verdict(entry):
    proof = read_proof(entry["file"])
    inputs = [(i, tuple(v)) for i, v in entry["inputs"]]
    cfg = ob.PcsConfig(entry["pow_bits"], entry["log_blowup_factor"], entry["log_last_layer_degree_bound"],
                       entry["n_queries"])
    acc, reason = ob.verify_batch([proof], cfg, inputs)
    if entry["expect"] == "ok":
        assert acc[0] == 1 and reason[0] == 0
        with pytest.raises(TypeError):  # the configuration is required: nothing is taken from the proof's header
            ob.verify_batch([proof], None, inputs)
    else:
        assert acc[0] == 0 and reason[0] != 0


def test_permutation_counts():
    # SURVEY App. C (transcript + batched Merkle); the survey's transcript figure includes
    # ceil(n_q/4) query draws where ceil(n_q/8) suffice, hence the small constant offsets.
    assert ob.perm_count(read_proof("recursive_proof_16_15.bin")) == 233 + 4230 - 2
    assert ob.perm_count(read_proof("small_proof.bin"), [(1, (1, 0, 0, 0))]) == 105 + 2736 - 2


def test_wrong_config_rejected():
    proof = read_proof("small_proof.bin")
    cfg = ob.PcsConfig(20, 5, 2, 15)
    acc, reason = ob.verify_batch([proof], cfg, [(1, (1, 0, 0, 0))])
    assert acc[0] == 0 and reason[0] == 1


def test_wrong_public_input_rejected():
    proof = read_proof("small_proof.bin")
    acc, reason = ob.verify_batch([proof], fixture_cfg("small_proof.bin"), [(1, (2, 0, 0, 0))])
    assert acc[0] == 0 and reason[0] == 3  # logup
    acc, reason = ob.verify_batch([read_proof("recursive_proof_16_15.bin")], fixture_cfg("recursive_proof_16_15.bin"), [(1, (1, 0, 0, 0))])
    assert acc[0] == 0 and reason[0] == 3


def test_tampered_proofs_rejected():
    proof = read_proof("small_proof.bin")
    bad = [ob.tamper(proof, i) for i in range(40)]
    acc, reason = ob.verify_batch(bad, fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc.sum() == 0
    assert set(reason.tolist()) <= set(range(1, 13))
    # several different stages must be hit (pow, trees, FRI first/inner, parse)
    assert len(set(reason.tolist())) >= 4


def test_truncated_and_empty():
    proof = read_proof("small_proof.bin")
    for cut in (0, 4, 60, 3580, len(proof) - 4):
        acc, reason = ob.verify_batch([proof[:cut]], fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
        assert acc[0] == 0 and reason[0] == 1
    acc, reason = ob.verify_batch([proof + b"\0\0\0\0"], fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc[0] == 0 and reason[0] == 1
    acc, _ = ob.verify_batch([], fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert len(acc) == 0


def test_qm31_field_identities():
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = rng.integers(1, P, 4, dtype=np.uint32)
        assert ob.qm31_mul(a, ob.qm31_inv(a)).tolist() == [1, 0, 0, 0]
    # u^2 = 2 + i  
    assert ob.qm31_mul([0, 0, 1, 0], [0, 0, 1, 0]).tolist() == [2, 1, 0, 0]
    assert ob.qm31_mul([0, 1, 0, 0], [0, 1, 0, 0]).tolist() == [P - 1, 0, 0, 0]


def test_domain_points_on_circle():
    for log in (5, 13, 22):
        for q in (0, 1, 2, (1 << log) - 1):
            x, y = ob.domain_point(log, q)
            assert (x * x + y * y) % P == 1
        # bit 0 of the query selects the conjugate (primitives/query/src/lib.rs:139-143)
        x0, y0 = ob.domain_point(log, 6)
        x1, y1 = ob.domain_point(log, 7)
        assert x0 == x1 and (y0 + y1) % P == 0


def test_trace_paths_verify_against_commitments():
    """the per-query paths cherry-picked from the batched walk each recompute the committed root
    (SinglePathMerkleProof::verify, components/hints/src/decommit.rs:22-42)."""
    import struct
    for name, inputs in (("small_proof.bin", [(1, (1, 0, 0, 0))]), ("recursive_proof_16_15.bin", ob.STANDARD_INPUTS)):
        proof = read_proof(name)
        t = ob.transcript_raw(proof)
        nq, M = int(t[2]), int(t[3])
        sib, pos, depth = ob.trace_paths(proof, nq, M, inputs)
        words = np.frombuffer(proof, dtype=np.uint32)
        lp, lq, blowup = int(words[0]), int(words[1]), int(words[11])
        A, B = lp + blowup, lq + blowup
        assert depth.tolist() == [max(A, B)] * 3 + [M]
        raw = t[40 + 4 * int(t[1]):40 + 4 * int(t[1]) + nq] & ((1 << M) - 1)
        for tr in range(4):
            assert pos[tr].tolist() == (raw >> (M - int(depth[tr]))).tolist()
        # tree 3 (composition, 8 columns at the leaves only): rebuild its roots with the generic path hasher
        d = int(depth[3])
        # columns come from queried_values[3]: one row of 8 words per query, in ascending query order;
        # locate it by walking the bincode prefixes (decommitments start at word 895)
        order = np.argsort(raw, kind="stable")
        pos_w = 895 + 2
        for _ in range(4):
            nh = int(words[pos_w]); pos_w += 2 + 8 * nh + 2
        pos_w += 2
        qvs = []
        for _ in range(4):
            nv = int(words[pos_w]); qvs.append(words[pos_w + 2:pos_w + 2 + nv]); pos_w += 2 + nv
        rows = qvs[3].reshape(-1, 8)
        cols = np.zeros((nq, 8), np.uint32)
        cols[order] = rows
        n_cols_at = [0] * d + [8]
        roots = ob.merkle_path_root(pos[3], sib[3][:, :d, :], cols, n_cols_at)
        commitment3 = words[17 + 24:17 + 32]
        assert all(r.tolist() == commitment3.tolist() for r in roots)


def _pair_path_root(query, depth, data_levels, sib, cols):
    """SinglePairMerkleProof::verify (components/hints/src/folding.rs:33-91) with the oracle's hasher."""
    z8 = np.zeros(8, np.uint32)
    def colcap(v4):
        return ob.poseidon2_permute(np.concatenate([v4, np.zeros(4, np.uint32), z8]))[0][8:]
    def leaf(v4):
        return ob.hash_node(None, v4)[0]
    c = 0
    self_h, sib_h = leaf(cols[c][:4]), leaf(cols[c][4:])
    c += 1
    for i in range(depth):
        h = depth - i - 1
        l, r = (self_h, sib_h) if ((query >> i) & 1) == 0 else (sib_h, self_h)
        if h not in data_levels:
            self_h = ob.hash_node((l, r), np.zeros((1, 0)))[0]
            if i != depth - 1:
                sib_h = sib[i]
        else:
            self_h = ob.hash_node((l, r), cols[c][:4])[0]
            sib_h = ob.poseidon2_permute(np.concatenate([sib[i], colcap(cols[c][4:])]))[0][:8]
            c += 1
    return self_h


def test_fri_paths_verify_against_commitments():
    """per-query pair paths of every FRI tree recompute the layer commitments."""
    for name, inputs in (("small_proof.bin", [(1, (1, 0, 0, 0))]), ("recursive_proof_16_15.bin", ob.STANDARD_INPUTS)):
        proof = read_proof(name)
        lay = ob.proof_layout(proof)
        t = ob.transcript_raw(proof)
        nq, M = int(t[2]), int(t[3])
        nt = 1 + lay["n_inner"]
        sib, cols = ob.fri_paths(proof, nq, M, nt, inputs)
        raw = t[40 + 4 * int(t[1]):40 + 4 * int(t[1]) + nq] & ((1 << M) - 1)
        A, B = lay["lp"] + lay["blowup"], lay["lq"] + lay["blowup"]
        for s2 in range(nt):
            depth = M if s2 == 0 else M - s2
            data_levels = {M, A, B} if s2 == 0 else {depth}
            for q in (0, nq // 2, nq - 1):
                query = int(raw[q]) >> (M - depth)
                root = _pair_path_root(query, depth, data_levels, sib[s2, q], cols[s2, q])
                assert root.tolist() == lay["fri_commitments"][s2].tolist(), (name, s2, q)


def test_every_length_prefix_mutated_is_rejected():
    """Every u64 length prefix of the variable part perturbed: the oracle must reject (and not crash)."""
    proof = read_proof("small_proof.bin")
    batch = []
    for pos, n, _ in ob.proof_layout(proof)["prefixes"]:
        for val in {max(n - 1, 0), n + 1, 0, 0xFFFFFFFF, (1 << 32) + n, 8 * n + 3} - {n}:
            b = bytearray(proof)
            b[4 * pos:4 * pos + 8] = int(val).to_bytes(8, "little")
            batch.append(bytes(b))
    acc, reason = ob.verify_batch(batch + [proof], fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc[-1] == 1 and int(acc[:-1].sum()) == 0
    assert set(reason[:-1].tolist()) <= set(range(1, 13))


def test_structural_mutants_fail_in_the_consuming_stage():
    """Re-serialized proofs with one witness list perturbed parse fine and fail where the list is consumed
    (reasons: 6..9 = trace tree t, 10 = FRI first layer, 11 = FRI inner layers, 1 = parser's shape rules)."""
    proof = read_proof("small_proof.bin")
    mut = ob.structural_mutants(proof)
    acc, reason = ob.verify_batch([b for _, b in mut], fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert int(acc.sum()) == 0
    for (tag, _), r in zip(mut, reason.tolist()):
        if tag.startswith(("hw[", "qv[")):
            assert r == 6 + int(tag[3]), (tag, r)
        elif tag.startswith("layer[0]"):
            assert r == 10, (tag, r)
        elif tag.startswith("layer["):
            assert r == 11, (tag, r)
        elif tag.startswith("last") or tag in ("inner layers -1", "inner layers +1"):
            assert r == 1, (tag, r)


@pytest.mark.parametrize("name", ["small_proof.bin", "recursive_proof_16_15.bin", "level2-1.bin", "level12-1.bin"])
def test_trace_paths_with_columns_recompute_all_four_roots(name):
    """SinglePathMerkleProof {query, sibling_hashes, columns} per query and tree -> verify() (decommit.rs:22-42):
    every path of every tree recomputes the tree's commitment."""
    entry = next(e for e in load_manifest() if e["file"] == name)
    inputs = [(i, tuple(v)) for i, v in entry["inputs"]]
    proof = read_proof(name)
    words = np.frombuffer(proof, dtype=np.uint32)
    lp, lq, blowup, nq = int(words[0]), int(words[1]), int(words[11]), int(words[13])
    A, B = lp + blowup, lq + blowup
    M = max(lp + 1, lq + 2) + blowup
    sib, pos, depth = ob.trace_paths(proof, nq, M, inputs)
    cols = ob.trace_cols(proof, inputs)
    plonk, poseidon = [10, 12, 8], [40, 48, 8]
    for t in range(4):
        d = int(depth[t])
        n_cols_at = [0] * (d + 1)
        if t == 3:
            n_cols_at[M] = 8
        else:
            n_cols_at[A] += plonk[t]
            n_cols_at[B] += poseidon[t]
        tot = sum(n_cols_at)
        roots = ob.merkle_path_root(pos[t], sib[t][:, :d, :], cols[t][:, :tot], n_cols_at)
        want = words[17 + 8 * t:25 + 8 * t]
        assert all(r.tolist() == want.tolist() for r in roots), t


def test_oracle_is_clean_under_sanitizers():
    """The oracle's C source built with AddressSanitizer + UBSan (oracle/Makefile `asan`) runs the mutant corpus
    (tests/mutants.py) of three fixtures without a report.  CPU only — GPU sanitizers are not available."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               RSV_ORACLE_LIB=os.path.join(root, "oracle", "librsv_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "tests.mutants", "small_proof.bin", "level12-1.bin", "level2-1.bin"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "librsv_oracle_asan.so" in out.stdout


P31 = (1 << 31) - 1


def _cm(a, b):  # (a0 + a1 i)(b0 + b1 i), i^2 = -1
    return ((a[0] * b[0] - a[1] * b[1]) % P31, (a[0] * b[1] + a[1] * b[0]) % P31)


def _qmul(x, y):  # (xa + xb u)(ya + yb u), u^2 = 2 + i   (primitives/fields/src/qm31.rs)
    xa, xb, ya, yb = x[:2], x[2:], y[:2], y[2:]
    bb = _cm(xb, yb)
    r = _cm(bb, (2, 1))
    aa = _cm(xa, ya)
    ab, ba = _cm(xa, yb), _cm(xb, ya)
    return ((aa[0] + r[0]) % P31, (aa[1] + r[1]) % P31, (ab[0] + ba[0]) % P31, (ab[1] + ba[1]) % P31)


def test_field_ops_match_python_model():
    """a1/a2: the oracle's M31 / CM31 / QM31 arithmetic against big-integer Python (products, inverses through
    x * x^-1 = 1, powers), including the edge values 0, 1 and P - 1."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, P31, (64, 4), dtype=np.uint32)
    b = rng.integers(0, P31, (64, 4), dtype=np.uint32)
    a[:4] = [[0, 0, 0, 0], [1, 0, 0, 0], [P31 - 1] * 4, [0, 0, 1, 0]]
    b[:4] = [[P31 - 1] * 4, [P31 - 1, 0, 0, 0], [P31 - 1] * 4, [0, 0, 1, 0]]
    mul = ob.field_op(2, a, b)
    for i in range(64):
        assert tuple(int(v) for v in mul[i]) == _qmul([int(v) for v in a[i]], [int(v) for v in b[i]])
    assert ob.field_op(2, [[0, 0, 1, 0]], [[0, 0, 1, 0]]).tolist() == [[2, 1, 0, 0]]  # u^2 = 2 + i
    add, sub = ob.field_op(0, a, b), ob.field_op(1, a, b)
    assert np.array_equal(add, ((a.astype(np.uint64) + b) % P31).astype(np.uint32))
    assert np.array_equal(sub, ((a.astype(np.uint64) + P31 - b) % P31).astype(np.uint32))
    inv = ob.field_op(3, a[1:])
    assert all(tuple(int(v) for v in r) == (1, 0, 0, 0) for r in ob.field_op(2, a[1:], inv))
    minv = ob.field_op(5, a[1:])
    assert all(int(x[0]) * int(y[0]) % P31 == 1 for x, y in zip(a[1:], minv) if x[0])
    cinv = ob.field_op(7, a[1:])
    assert all(_cm((int(x[0]), int(x[1])), (int(y[0]), int(y[1]))) == (1, 0) for x, y in zip(a[1:], cinv) if x[0] or x[1])
    assert np.array_equal(ob.field_op(8, a), ob.field_op(2, a, np.tile([0, 1, 0, 0], (64, 1))))
    assert np.array_equal(ob.field_op(9, a), ob.field_op(2, a, np.tile([0, 0, 1, 0], (64, 1))))
    e = np.zeros((64, 4), np.uint32)
    e[:, 0] = rng.integers(0, 1 << 32, 64, dtype=np.uint64).astype(np.uint32)
    e[:3, 0] = [0, 1, 5]
    pw = ob.field_op(10, a, e)
    for i in range(8):
        acc, base, k = (1, 0, 0, 0), [int(v) for v in a[i]], int(e[i, 0])
        while k:
            if k & 1:
                acc = _qmul(acc, base)
            base = _qmul(base, base)
            k >>= 1
        assert tuple(int(v) for v in pw[i]) == tuple(acc)


def test_line_eval_matches_fold_definition():
    """primitives/line/src/lib.rs:39-67 restated with Python integers: fold(coeffs, [x, pi(x), ...])."""
    rng = np.random.default_rng(4)
    for log_n in (0, 1, 2, 4, 8):
        coeffs = rng.integers(0, P31, (1 << log_n, 4), dtype=np.uint32)
        xs = rng.integers(0, P31, 5, dtype=np.uint32)
        got = ob.line_eval(coeffs, xs)
        for t, x in enumerate(int(v) for v in xs):
            d = []
            for _ in range(log_n):
                d.append(x)
                x = (2 * x * x - 1) % P31

            def fold(vals, f):
                if len(vals) == 1:
                    return [int(v) for v in vals[0]]
                half = len(vals) // 2
                lo, hi = fold(vals[:half], f[1:]), fold(vals[half:], f[1:])
                return [(lo[k] + hi[k] * f[0]) % P31 for k in range(4)]
            assert got[t].tolist() == fold(list(coeffs), d)


def test_reject_fixtures_behind_the_proof_of_work():
    """tests/golden/make_reject_fixtures.py: small_proof.bin with a changed sampled value / with a duplicate query
    position, nonce re-ground so that the proof-of-work still passes -> RSV_R_COMPOSITION (4) and RSV_R_DUP_QUERY (5).
    The un-reground variant of the first one stops at the proof of work (2)."""
    inputs = [(1, (1, 0, 0, 0))]
    comp, dup, good = read_proof("small_proof_composition.bin"), read_proof("small_proof_dup_query.bin"), read_proof("small_proof.bin")
    lay = ob.proof_layout(good)
    stale = bytearray(comp)
    stale[4 * lay["nonce_word"]:4 * lay["nonce_word"] + 8] = good[4 * lay["nonce_word"]:4 * lay["nonce_word"] + 8]
    acc, reason = ob.verify_batch([comp, dup, good, bytes(stale)], fixture_cfg("small_proof.bin"), inputs)
    assert acc.tolist() == [0, 0, 1, 0] and reason.tolist() == [4, 5, 0, 2]
    # the duplicate really is one: the masked query positions of the re-ground transcript collide
    t = ob.transcript_raw(dup)
    q = t[40 + 4 * int(t[1]):40 + 4 * int(t[1]) + 16] & ((1 << int(t[3])) - 1)
    assert len(set(q.tolist())) < 16 and t[0] == 0


def _header_edit(proof, word, value):
    b = bytearray(proof)
    b[4 * word:4 * word + 4] = int(value).to_bytes(4, "little")
    return bytes(b)


def security_downgrade_batch():
    """Proofs whose SERIALIZED configuration words were lowered by a forger (pow_bits 20 -> 0 / 1, n_queries 16 -> 1,
    log_blowup 5 -> 1, log_last 2 -> 0): the verifier must not take its security level from them."""
    proof = read_proof("small_proof.bin")
    edits = [(10, 0), (10, 1), (10, 19), (13, 1), (13, 15), (11, 1), (11, 4), (12, 0), (12, 3)]
    return [_header_edit(proof, w, v) for w, v in edits] + [_header_edit(_header_edit(proof, 13, 1), 11, 1), proof]


def test_lowered_security_words_are_rejected():
    """pow_bits / n_queries / blowup words lowered in the proof header.  Under the caller's
    configuration every such proof is RSV_R_PARSE; pow_bits = 0 in the header must not turn the PoW check off."""
    batch = security_downgrade_batch()
    acc, reason = ob.verify_batch(batch, fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc.tolist() == [0] * (len(batch) - 1) + [1]
    assert reason.tolist() == [1] * (len(batch) - 1) + [0]
    # a per-proof configuration index outside the table rejects that proof only
    cs, keep = ob.make_cfg_set([fixture_cfg("small_proof.bin"), fixture_cfg("level1-5.bin")], 2)
    import ctypes
    of = np.array([0, 7], np.uint8)
    cs.cfg_of = of.ctypes.data
    good = read_proof("small_proof.bin")
    blob, offsets = ob.pack([good, good])
    acc, reason = np.zeros(2, np.uint8), np.zeros(2, np.uint8)
    pi = ob.make_inputs([(1, (1, 0, 0, 0))])
    rc = ob.lib.rsvo_verify_batch(blob.ctypes.data_as(ob._u8p), offsets.ctypes.data_as(ob._u64p), 2, ctypes.byref(cs), pi, 1,
                                  acc.ctypes.data_as(ob._u8p), reason.ctypes.data_as(ob._u8p))
    assert rc == 0 and acc.tolist() == [1, 0] and reason.tolist() == [0, 1]
    # no configuration at all is API misuse, not "trust the header"
    assert ob.lib.rsvo_verify_batch(blob.ctypes.data_as(ob._u8p), offsets.ctypes.data_as(ob._u64p), 2, None, pi, 1,
                                    acc.ctypes.data_as(ob._u8p), reason.ctypes.data_as(ob._u8p)) == -1


def oods_params_of(proof):
    """The 26 parameter words of rsv_oods_eval / rsvo_oods_eval for a genuine proof, from the oracle's transcript."""
    w = np.frombuffer(proof, dtype=np.uint32)
    t = ob.transcript_raw(proof)
    return np.concatenate([w[0:2], w[2:10], t[4:8], t[8:12], t[12:16], t[20:24]]).astype(np.uint32)


@pytest.mark.parametrize("entry", [e for e in load_manifest() if e["expect"] == "ok"], ids=lambda e: e["file"])
def test_oods_eval_probe_on_fixtures(entry):
    """Row a10 on its own: on every accepting fixture the 86-constraint accumulator equals the value of the committed
    composition polynomial (that IS the reference's check, composition/src/lib.rs:106-120); one changed sample or
    challenge breaks the equality."""
    proof = read_proof(entry["file"])
    sm, pr = ob.sampled_values(proof), oods_params_of(proof)
    out = ob.oods_eval(sm[None], pr[None])[0]
    assert out[:4].tolist() == out[4:].tolist() and any(out[:4])
    for k in (0, 57, 110, 133):
        bad = sm.copy(); bad[k, 1] = (int(bad[k, 1]) + 1) % 0x7FFFFFFF
        o = ob.oods_eval(bad[None], pr[None])[0]
        assert o[:4].tolist() != o[4:].tolist(), k
    for k in (10, 14, 18, 22):  # z, alpha, random_coeff, oods.x
        badp = pr.copy(); badp[k] = (int(badp[k]) + 1) % 0x7FFFFFFF
        o = ob.oods_eval(sm[None], badp[None])[0]
        assert o[:4].tolist() != o[4:].tolist(), k


def test_query_dump_is_consistent_with_the_older_probes():
    proof = read_proof("recursive_proof_16_15.bin")
    d = ob.query_dump(proof)
    ni = d.shape[1] // 4 - 8
    assert np.array_equal(d[:, 12:24].reshape(16, 3, 4).transpose(1, 0, 2), ob.fri_folded(proof))
    assert np.array_equal(d[:, 24 + 4 * ni:28 + 4 * ni], d[:, 28 + 4 * ni:32 + 4 * ni])  # folded == last-layer evaluation


def test_big_shape_reject_fixtures_behind_the_proof_of_work():
    """The same two rejection stages for big shapes (tests/golden/make_reject_fixtures.py big): a 2^16 / 2^15 proof with
    one changed sampled value, and the 80-query level1-5 proof re-ground until two of its positions collide."""
    comp, dup = read_proof("recursive_proof_16_15_composition.bin"), read_proof("level1-5_dup_query.bin")
    acc, reason = ob.verify_batch([comp, read_proof("recursive_proof_16_15.bin"), dup, read_proof("level1-5.bin")],
                                  [fixture_cfg("recursive_proof_16_15.bin")] * 2 + [fixture_cfg("level1-5.bin")] * 2)
    assert acc.tolist() == [0, 1, 0, 1] and reason.tolist() == [4, 0, 5, 0]
    t = ob.transcript_raw(dup)
    q = t[40 + 4 * int(t[1]):40 + 4 * int(t[1]) + 80] & ((1 << int(t[3])) - 1)
    assert len(set(q.tolist())) < 80 and t[0] == 0


# ---------------------------------------------------------------------------------------------- emulated Poseidon2
EMULATED_KAT = [260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943, 1682438524,
                1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264]


def test_emulated_poseidon_reference_test():
    """primitives/poseidon31/src/emulated.rs:236-275 replayed on the oracle's constraint system: 16 witness words
    0..15 packed four to a QM31, permuted with is_swap None, Some((false, 0)) and — halves exchanged —
    Some((true, 1)); every result is the known-answer state and every row passes check_arithmetics."""
    cs = ob.EmulatedCS()
    m = [cs.witness_m31(i) for i in range(16)]
    left = [cs.qm31_from_m31(m[0:4]), cs.qm31_from_m31(m[4:8])]
    right = [cs.qm31_from_m31(m[8:12]), cs.qm31_from_m31(m[12:16])]
    outs = [cs.permute(left, right, None), cs.permute(left, right, 0), cs.permute(right, left, 1)]
    v, kind, rows = cs.export()
    for o in outs:
        assert v[o].reshape(-1).tolist() == EMULATED_KAT
    assert cs.check_arithmetics() == 0
    assert len(rows) == cs.n_rows and len(v) == cs.n_vars
    # every row is one of the six gate types of plonk_without_poseidon.rs:478-560
    ops = {tuple(r[4:7]) for r in rows}
    assert ops <= {(0, 0, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 0), (0, 1, 1)}
    # a corrupted variable is caught by the gate equations
    bad = v[outs[0][0]].copy()
    bad[0] ^= 1
    cs.set_vars(int(outs[0][0]), bad)
    assert cs.check_arithmetics() != 0


def test_emulated_poseidon_matches_plain_permutation():
    """The gadget's outputs are poseidon2_permute of the (possibly exchanged) halves; constants are allocated once;
    a steady-state call appends 401 variables (413 with a swap bit), as include/rsv.h states."""
    rng = np.random.default_rng(11)
    n = 24
    left = rng.integers(0, P, (n, 8), dtype=np.uint32)
    right = rng.integers(0, P, (n, 8), dtype=np.uint32)
    swap = rng.integers(0, 3, n).astype(np.uint8)
    left[0], right[0] = 0, P - 1  # extremes
    rows, cs, spans = ob.emulated_rows(left, right, swap)
    state = np.where((swap == 2)[:, None], np.concatenate([right, left], 1), np.concatenate([left, right], 1))
    want = ob.poseidon2_permute(state)
    assert (rows[:, 409:413, :].reshape(n, 16) == want).all() and not rows[:, 413:].any()
    assert (rows[swap == 0, :12] == 0).all()


# ------------------------------------------------------------------------------------------------ PoseidonFlow
# (source proof, how many times the circuit verifies it, the proof of that circuit): components/test_data +
# examples/single-proof/src/main.rs:23-103, examples/multi-proofs/src/main.rs:198-295
FLOW_CHAIN = [("small_proof.bin", 1, "recursive_proof_16_15.bin"), ("recursive_proof_16_15.bin", 5, "level1-5.bin"),
              ("level1-5.bin", 1, "level2-1.bin"), ("level2-1.bin", 1, "level3-1.bin"), ("level3-1.bin", 5, "level4-5.bin"),
              ("level4-5.bin", 1, "level5-1.bin"), ("level5-1.bin", 1, "level6-1.bin"), ("level6-1.bin", 1, "level7-1.bin"),
              ("level7-1.bin", 1, "level8-1.bin"), ("level8-1.bin", 1, "level9-1.bin"), ("level9-1.bin", 1, "level10-1.bin"),
              ("level10-1.bin", 1, "level11-1.bin"), ("level11-1.bin", 1, "level12-1.bin"), ("level12-1.bin", 1, "level13-1.bin"),
              ("level13-1.bin", 1, "hybrid_hash.bin")]


@pytest.mark.parametrize("src,mult,dst", FLOW_CHAIN, ids=[c[0] for c in FLOW_CHAIN])
def test_poseidon_flow_count_predicts_next_level(src, mult, dst):
    """The one thing the reference pins about the flow: the circuit that verifies `src` `mult` times records
    mult x count Poseidon invocations, pads them (plonk_with_poseidon.rs:282-300) and proves them as the Poseidon
    component of `dst`, whose log size is the second word of that fixture.  The oracle's invocation count must land on
    it for every consecutive pair of the reference's fixtures (three of them within 4 % of a power-of-two boundary)."""
    man = {e["file"]: e for e in load_manifest()}
    e = man[src]
    flow = ob.poseidon_flow(read_proof(src), [(i, tuple(v)) for i, v in e["inputs"]])
    want = int(np.frombuffer(read_proof(dst)[:8], np.uint32)[1])
    assert want == man[dst]["log_size_poseidon"]
    assert ob.flow_log_size(mult * len(flow)) == want, (len(flow), mult, want)


@pytest.mark.parametrize("name", ["small_proof.bin", "recursive_proof_16_15.bin", "level2-1.bin", "level1-5.bin", "level13-1.bin"])
def test_poseidon_flow_records_are_invocations(name):
    """Every record is a Poseidon2HalfVar::permute invocation: perm(swap ? right || left : left || right) = rate || cap
    (what check_poseidon_invocations asserts, plonk_with_poseidon.rs:468-519); the transcript part is the channel chain
    (each mix's capacity is the next record's right half); the flow ends in the paths' roots = the commitments."""
    e = next(x for x in load_manifest() if x["file"] == name)
    proof = read_proof(name)
    flow = ob.poseidon_flow(proof, [(i, tuple(v)) for i, v in e["inputs"]])
    left, right, out, swap = flow[:, 0:8], flow[:, 8:16], flow[:, 16:32], flow[:, 32]
    assert set(swap.tolist()) <= {0, 1} and (flow[:, :32] < 0x7FFFFFFF).all()
    state = np.where(swap[:, None] == 1, np.concatenate([right, left], 1), np.concatenate([left, right], 1))
    assert np.array_equal(ob.poseidon2_permute(state), out)
    words = np.frombuffer(proof, np.uint32)
    # record 0 = mix_root(commitment 0) on the zero digest; record 1 = mix_one_felt(lp) on its capacity
    assert left[0].tolist() == words[17:25].tolist() and not right[0].any()
    assert right[1].tolist() == out[0, 8:16].tolist() and left[1].tolist() == [e["log_size_plonk"]] + [0] * 7
    # the first trace-tree path ends in commitment 0: its last record's rate
    nq, lay = e["n_queries"], ob.proof_layout(proof)
    n_transcript = 4 + 1 + 2 + 1 + 1 + 1 + 71 + 1 + 2 * (1 + lay["n_inner"]) + (1 << e["log_last_layer_degree_bound"]) // 2 + (
        1 if e["log_last_layer_degree_bound"] == 0 else 0) + 1 + (nq + 3) // 4
    A, B = e["log_size_plonk"] + e["log_blowup_factor"], e["log_size_poseidon"] + e["log_blowup_factor"]
    lo, hi = (10, 40) if A < B else (40, 10)  # tree 0: columns at the lower / the leaf level
    per_path0 = -(-hi // 8) + 1 + max(A, B) + (0 if A == B else -(-lo // 8) + 1)
    if A == B:
        per_path0 = -(-50 // 8) + 1 + A
    assert out[n_transcript + per_path0 - 1, :8].tolist() == words[17:25].tolist()
    assert swap[:n_transcript].sum() == 0
