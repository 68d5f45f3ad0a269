"""GPU parity tests: every C-ABI entry point of librsv_hip.so against the CPU oracle on the same
inputs (bit-exact: all arithmetic is integer), and against the committed golden fixtures."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.conftest import fixture_cfg, load_manifest, read_proof

pytestmark = pytest.mark.gpu
P = 0x7FFFFFFF
KAT_OUT = [260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943,
           1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264]


def entry_inputs(e):
    return [(i, tuple(v)) for i, v in e["inputs"]]


def test_device_present(rsv):
    assert rsv.device_count() >= 1


def test_poseidon2_kat(rsv):
    assert rsv.poseidon2_permute(np.arange(16, dtype=np.uint32))[0].tolist() == KAT_OUT


def test_poseidon2_random_states(rsv):
    rng = np.random.default_rng(1)
    n = 1 << 15
    s = rng.integers(0, P, (n, 16), dtype=np.uint32)
    s[0] = 0
    s[1] = P - 1
    assert np.array_equal(rsv.poseidon2_permute(s), ob.poseidon2_permute(s))


def test_poseidon2_ragged_sizes(rsv):
    rng = np.random.default_rng(2)
    for n in (1, 63, 64, 65, 255, 257, 1000):
        s = rng.integers(0, P, (n, 16), dtype=np.uint32)
        assert np.array_equal(rsv.poseidon2_permute(s), ob.poseidon2_permute(s))
    assert rsv.poseidon2_permute(np.zeros((0, 16), np.uint32)).shape == (0, 16)


def test_poseidon2_rejects_noncanonical(rsv):
    s = np.zeros((4, 16), np.uint32)
    s[2, 5] = P
    with pytest.raises(rsv.RsvError) as e:
        rsv.poseidon2_permute(s)
    assert e.value.code == -5


def test_half_permute(rsv):
    rng = np.random.default_rng(3)
    n = 1000
    l = rng.integers(0, P, (n, 8), dtype=np.uint32)
    r = rng.integers(0, P, (n, 8), dtype=np.uint32)
    sw = rng.integers(0, 2, n, dtype=np.uint8)
    for swap in (None, sw):
        rate, cap = rsv.half_permute(l, r, swap)
        orate, ocap = ob.half_permute(l, r, swap)
        assert np.array_equal(rate, orate) and np.array_equal(cap, ocap)


def test_half_permute_known_answer_with_swap(rsv):
    """primitives/poseidon31/src/emulated.rs:236-275: the KAT through Poseidon2HalfVar::permute, three swap variants."""
    kat = [260776483, 1182896747, 1656699352, 746018898, 102875940, 1812541025, 515874083, 755063943, 1682438524, 1265420601, 238640995, 200799880, 1659717477, 2080202267, 1269806256, 1287849264]
    lo, hi = np.arange(8, dtype=np.uint32)[None], np.arange(8, 16, dtype=np.uint32)[None]
    for l, r, sw in ((lo, hi, None), (lo, hi, np.array([0], np.uint8)), (hi, lo, np.array([1], np.uint8))):
        rate, cap = rsv.half_permute(l, r, sw)
        assert np.concatenate([rate[0], cap[0]]).tolist() == kat


def test_last_layer_log_size_word_is_not_a_field_element(rsv):
    """The proof's final word, last_layer_poly.log_size, is never read by the reference (it takes the size from
    coeffs.len(), components/hints/src/folding.rs:573): any u32 there, canonical or not, leaves the verdict alone.
    (Found by the extended soak: the canonicity scan used to call such a proof PARSE.)"""
    inputs_of = {e["file"]: entry_inputs(e) for e in load_manifest()}
    for name in ("small_proof.bin", "recursive_proof_16_15.bin"):
        proof = read_proof(name)
        cfg = fixture_cfg(name)
        batch = []
        for val in (0, 9, P - 1, P, 0x80000008, 0xFFFFFFFF):
            w = np.frombuffer(proof, np.uint32).copy()
            w[-1] = val
            batch.append(w.tobytes())
        # ... while the word before it (the last coefficient) is a field element
        w = np.frombuffer(proof, np.uint32).copy()
        w[-2] = P
        batch.append(w.tobytes())
        acc, reason = rsv.verify_batch(batch, cfg, inputs_of[name])
        oacc, oreason = ob.verify_batch(batch, cfg, inputs_of[name])
        assert acc.tolist() == oacc.tolist() == [1] * 6 + [0]
        assert reason.tolist() == oreason.tolist() and reason[-1] == 1


# ---------------------------------------------------------------------------------------------- f4: emulated Poseidon2
def test_emulated_reference_test_on_gpu(rsv):
    """primitives/poseidon31/src/emulated.rs:236-275 with the GPU's rows: the three permutes of the reference's test
    give the known answer, and — written over the oracle constraint system's variables — every GPU value satisfies
    the gate equations of check_arithmetics (constraint_system/src/plonk_without_poseidon.rs:410-598)."""
    lo, hi = np.arange(8, dtype=np.uint32), np.arange(8, 16, dtype=np.uint32)
    left, right, swap = np.stack([lo, lo, hi]), np.stack([hi, hi, lo]), np.array([0, 1, 2], np.uint8)
    rows = rsv.poseidon2_emulated(left, right, swap)
    assert rows.shape == (3, 416, 4)
    for p in range(3):
        assert rows[p, 409:413].reshape(-1).tolist() == KAT_OUT
    want, cs, spans = ob.emulated_rows(left, right, swap)
    assert np.array_equal(rows, want)
    for p, (a, b, sw) in enumerate(spans):
        cs.set_vars(a, rows[p, (0 if sw else 12):413])
    assert cs.check_arithmetics() == 0
    # and the equations do bite: one flipped GPU word fails a row
    bad = rows[1, 40].copy()
    bad[2] ^= 1
    cs.set_vars(spans[1][0] + 40, bad)
    assert cs.check_arithmetics() != 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 200])
def test_emulated_rows_match_oracle(rsv, n):
    rng = np.random.default_rng(100 + n)
    left = rng.integers(0, P, (n, 8), dtype=np.uint32)
    right = rng.integers(0, P, (n, 8), dtype=np.uint32)
    swap = rng.integers(0, 3, n).astype(np.uint8)
    left[0], right[0] = P - 1, 0
    want, _, _ = ob.emulated_rows(left, right, swap)
    assert np.array_equal(rsv.poseidon2_emulated(left, right, swap), want)
    want0, _, _ = ob.emulated_rows(left, right, np.zeros(n, np.uint8))
    assert np.array_equal(rsv.poseidon2_emulated(left, right, None), want0)
    assert rsv.poseidon2_emulated(np.zeros((0, 8), np.uint32), np.zeros((0, 8), np.uint32)).shape == (0, 416, 4)


def test_emulated_rejects_bad_inputs(rsv):
    z = np.zeros((70, 8), np.uint32)
    bad = z.copy()
    bad[66, 3] = P
    for l, r, sw in ((bad, z, None), (z, bad, None), (z, z, np.array([0] * 69 + [3], np.uint8))):
        with pytest.raises(rsv.RsvError) as e:
            rsv.poseidon2_emulated(l, r, sw)
        assert e.value.code == -5


def test_emulated_dev_large_batch_properties(rsv):
    """2^16 permutations resident in HBM (436 MB of rows): the output rows are the plain permutation of the (possibly
    exchanged) halves, the None-mode swap rows are zero, the structure of a partial round holds everywhere (row 9 of
    round r is the sum of rows 7 and 8; rows 1-2 split limb 0), and a sample of permutations equals the oracle."""
    import torch
    n = 1 << 16
    rng = np.random.default_rng(77)
    left = rng.integers(0, P, (n, 8), dtype=np.uint32)
    right = rng.integers(0, P, (n, 8), dtype=np.uint32)
    swap = rng.integers(0, 3, n).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_l, d_r = torch.from_numpy(left.view(np.int32)).to(dev), torch.from_numpy(right.view(np.int32)).to(dev)
    d_s = torch.from_numpy(swap).to(dev)
    d_rows = torch.empty((n, 416, 4), dtype=torch.int32, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = rsv.Context(0)
    ctx.poseidon2_emulated(d_l, d_r, d_s, d_rows, d_bad)
    ctx.synchronize()
    assert int(d_bad.item()) == 0
    rows = d_rows.cpu().numpy().view(np.uint32)
    state = np.where((swap == 2)[:, None], np.concatenate([right, left], 1), np.concatenate([left, right], 1))
    assert np.array_equal(rows[:, 409:413].reshape(n, 16), rsv.poseidon2_permute(state))
    assert not rows[swap == 0, :12].any() and not rows[:, 413:].any()
    first_partial = 12 + 11 + 4 * 19
    for r in (0, 13):
        b = first_partial + 17 * r
        s = (rows[:, b + 6].astype(np.uint64) + rows[:, b + 7]) % P
        assert np.array_equal(s, rows[:, b + 8])
        assert not rows[:, b, 1:].any() and not rows[:, b + 1, 0].any()
    pick = rng.choice(n, 64, replace=False)
    want, _, _ = ob.emulated_rows(left[pick], right[pick], swap[pick])
    assert np.array_equal(rows[pick], want)
    ctx.close()


@pytest.mark.parametrize("n_cols", [0, 1, 4, 7, 8, 13, 16, 17, 21, 25, 48])
def test_hash_node(rsv, n_cols):
    # column lengths 7/13/16/17/21/25 are the ones primitives/merkle/src/lib.rs:207-303 tests
    rng = np.random.default_rng(n_cols)
    n = 300
    cols = rng.integers(0, P, (n, n_cols), dtype=np.uint32)
    l = rng.integers(0, P, (n, 8), dtype=np.uint32)
    r = rng.integers(0, P, (n, 8), dtype=np.uint32)
    assert np.array_equal(rsv.hash_node((l, r), cols), ob.hash_node((l, r), cols))
    if n_cols:
        assert np.array_equal(rsv.hash_node(None, cols), ob.hash_node(None, cols))


def test_hash_node_checkpoints(rsv):
    assert rsv.hash_node(None, [1, 2, 3, 4, 5])[0].tolist() == [
        557709851, 1113733662, 222169927, 1376019790, 387901840, 1087892516, 628125718, 969660801]


def test_merkle_path_root(rsv):
    rng = np.random.default_rng(5)
    depth = 13
    n_cols_at = [0] * (depth + 1)
    n_cols_at[depth] = 40
    n_cols_at[9] = 10
    n_cols_at[4] = 3
    n = 200
    q = rng.integers(0, 1 << depth, n, dtype=np.uint32)
    sib = rng.integers(0, P, (n, depth, 8), dtype=np.uint32)
    cols = rng.integers(0, P, (n, sum(n_cols_at)), dtype=np.uint32)
    assert np.array_equal(rsv.merkle_path_root(q, sib, cols, n_cols_at), ob.merkle_path_root(q, sib, cols, n_cols_at))


@pytest.mark.parametrize("entry", load_manifest(), ids=lambda e: e["file"])
def test_transcript_matches_oracle(rsv, entry):
    proof = read_proof(entry["file"])
    want = rsv._parse_transcript(ob.transcript_raw(proof))
    got = rsv.transcript(proof)
    assert got == want


def test_all_fixtures_one_mixed_batch(rsv, manifest):
    std = [e for e in manifest if len(e["inputs"]) == 3]
    proofs = [read_proof(e["file"]) for e in std]
    cfgs = [fixture_cfg(e["file"]) for e in std]
    acc, reason = rsv.verify_batch(proofs, cfgs)
    oacc, oreason = ob.verify_batch(proofs, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    for e, a in zip(std, acc):
        assert bool(a) == (e["expect"] == "ok"), e["file"]


@pytest.mark.parametrize("entry", load_manifest(), ids=lambda e: e["file"])
def test_fixture_verdict(rsv, entry):
    proof = read_proof(entry["file"])
    cfg = rsv.PcsConfig(entry["pow_bits"], entry["log_blowup_factor"], entry["log_last_layer_degree_bound"],
                        entry["n_queries"])
    acc, reason = rsv.verify_batch([proof], cfg, entry_inputs(entry))
    assert bool(acc[0]) == (entry["expect"] == "ok")
    ocfg = ob.PcsConfig(entry["pow_bits"], entry["log_blowup_factor"], entry["log_last_layer_degree_bound"],
                        entry["n_queries"])
    oacc, oreason = ob.verify_batch([proof], ocfg, entry_inputs(entry))
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()


@pytest.mark.parametrize("name,n_tamper", [("small_proof.bin", 96), ("recursive_proof_16_15.bin", 64),
                                           ("level13-1.bin", 32), ("level2-1.bin", 16), ("level1-5.bin", 8)])
def test_tampered_proofs_match_oracle(rsv, manifest, name, n_tamper):
    entry = next(e for e in manifest if e["file"] == name)
    proof = read_proof(name)
    batch = [ob.tamper(proof, i) for i in range(n_tamper)] + [proof]
    acc, reason = rsv.verify_batch(batch, fixture_cfg(name), entry_inputs(entry))
    oacc, oreason = ob.verify_batch(batch, fixture_cfg(name), entry_inputs(entry))
    assert acc.tolist() == oacc.tolist()
    assert reason.tolist() == oreason.tolist()
    assert acc[-1] == 1 and acc[:-1].sum() == 0


def test_wrong_inputs_and_config(rsv):
    proof = read_proof("small_proof.bin")
    cfg = fixture_cfg("small_proof.bin")
    acc, reason = rsv.verify_batch([proof], cfg, [(1, (2, 0, 0, 0))])
    assert (acc[0], reason[0]) == (0, 3)
    acc, reason = rsv.verify_batch([proof], rsv.PcsConfig(20, 5, 2, 15), [(1, (1, 0, 0, 0))])
    assert (acc[0], reason[0]) == (0, 1)
    with pytest.raises(rsv.RsvError) as e:
        rsv.verify_batch([proof], cfg, [(1, (P, 0, 0, 0))])
    assert e.value.code == -5


def test_lowered_security_words_are_rejected(rsv):
    """a forger lowers pow_bits / n_queries / blowup / log_last in the proof header.  The GPU path
    takes the configuration from the caller only: every such proof is RSV_R_PARSE, as in the oracle; and no
    configuration at all is RSV_E_NULL on every verdict-producing entry point."""
    import ctypes
    from tests.test_oracle import security_downgrade_batch
    batch = security_downgrade_batch()
    cfg, inputs = fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))]
    acc, reason = rsv.verify_batch(batch, cfg, inputs)
    oacc, oreason = ob.verify_batch(batch, cfg, inputs)
    assert acc.tolist() == oacc.tolist() == [0] * (len(batch) - 1) + [1]
    assert reason.tolist() == oreason.tolist() == [1] * (len(batch) - 1) + [0]
    # per-proof configurations: the right one accepts, another fixture's rejects, an index beyond the table rejects
    good = read_proof("small_proof.bin")
    acc, reason = rsv.verify_batch([good, good], [cfg, fixture_cfg("level1-5.bin")], inputs)
    assert acc.tolist() == [1, 0] and reason.tolist() == [0, 1]
    pc = rsv.PreparedCfg([cfg, fixture_cfg("level1-5.bin")], np.array([0, 9], np.uint8))
    acc, reason = rsv.verify_batch([good, good], pc, inputs)
    assert acc.tolist() == [1, 0] and reason.tolist() == [0, 1]
    # NULL configuration: API misuse on the raw C-ABI
    blob, offsets = rsv.pack([good])
    a, r = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
    pi = rsv.make_inputs(inputs)
    u8, u64 = rsv._u8p, rsv._u64p
    assert rsv.lib.rsv_verify_batch(blob.ctypes.data_as(u8), offsets.ctypes.data_as(u64), 1, None, pi, 1, a.ctypes.data_as(u8),
                                    r.ctypes.data_as(u8), 0) == -1
    ho = rsv.HintsOut()
    assert rsv.lib.rsv_verify_hints(blob.ctypes.data_as(u8), offsets.ctypes.data_as(u64), 1, None, pi, 1, ctypes.byref(ho),
                                    a.ctypes.data_as(u8), r.ctypes.data_as(u8), 0) == -1
    out = np.zeros(rsv.TRANSCRIPT_WORDS, np.uint32)
    assert rsv.lib.rsv_transcript_batch(blob.ctypes.data_as(u8), offsets.ctypes.data_as(u64), 1, None, out.ctypes.data_as(rsv._u32p), 0) == -1
    ctx = rsv.Context(0)
    with pytest.raises(TypeError):
        ctx.verify_batch_host([good], None)
    ptrs = (ctypes.c_void_p * 1)(blob.ctypes.data)
    lens = np.array([blob.size], np.uint64)
    assert rsv.lib.rsv_verify_batch_host(ctx._h, ptrs, lens.ctypes.data_as(u64), 1, None, pi, 1, a.ctypes.data_as(u8), r.ctypes.data_as(u8)) == -1
    assert rsv.lib.rsv_verify_batch_dev(ctx._h, 16, 16, 1, None, pi, 1, 16, 16) == -1
    ctx.close()


def test_truncated_garbage_and_empty(rsv):
    proof = read_proof("small_proof.bin")
    rng = np.random.default_rng(7)
    batch = [proof[:cut] for cut in (0, 4, 60, 3580, len(proof) - 4)] + [proof + b"\0\0\0\0"]
    batch += [rng.integers(0, 256, 4096, dtype=np.uint8).tobytes(), proof]
    cfg = fixture_cfg("small_proof.bin")
    acc, reason = rsv.verify_batch(batch, cfg, [(1, (1, 0, 0, 0))])
    oacc, oreason = ob.verify_batch(batch, cfg, [(1, (1, 0, 0, 0))])
    assert acc.tolist() == oacc.tolist() == [0] * 7 + [1]
    assert reason.tolist() == oreason.tolist()
    acc, _ = rsv.verify_batch([], cfg, [(1, (1, 0, 0, 0))])
    assert len(acc) == 0


def test_config2_batch_1024_copies(rsv):
    """BASELINE config 2: 1 024 copies of recursive_proof_16_15.bin with the seeded tamper rule of
    SURVEY §8d (proof i with i % 17 == 5 gets one flipped bit); verdicts bit-exact vs the oracle."""
    proof = read_proof("recursive_proof_16_15.bin")
    batch = [ob.tamper(proof, i) if i % 17 == 5 else proof for i in range(1024)]
    cfg = fixture_cfg("recursive_proof_16_15.bin")
    acc, reason = rsv.verify_batch(batch, cfg)
    # the oracle only needs to judge the distinct inputs
    tampered = [i for i in range(1024) if i % 17 == 5]
    oacc, oreason = ob.verify_batch([batch[i] for i in tampered] + [proof], cfg)
    want_acc = np.full(1024, oacc[-1], np.uint8)
    want_reason = np.full(1024, oreason[-1], np.uint8)
    want_acc[tampered] = oacc[:-1]
    want_reason[tampered] = oreason[:-1]
    assert acc.tolist() == want_acc.tolist()
    assert reason.tolist() == want_reason.tolist()
    assert int(acc.sum()) == 1024 - len(tampered)


def test_device_resident_api_and_bitmap(rsv):
    import torch
    proof = read_proof("recursive_proof_16_15.bin")
    n = 300
    batch = [ob.tamper(proof, i) if i % 7 == 3 else proof for i in range(n)]
    blob, offsets = rsv.pack(batch)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    ctx = rsv.Context(0)
    assert all(v == 0 for v in ctx.last_stage_times().values())   # the stage clock is opt-in
    ctx.set_option("stage_times", "on")
    for _ in range(2):  # second call reuses the workspace
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=fixture_cfg("recursive_proof_16_15.bin"))
    ctx.accept_bitmap(d_acc, n, d_bitmap, d_count)
    ctx.synchronize()
    acc = d_acc.cpu().numpy()
    want = np.array([0 if i % 7 == 3 else 1 for i in range(n)], np.uint8)
    assert acc.tolist() == want.tolist()
    bits = np.unpackbits(d_bitmap.cpu().numpy().view(np.uint8), bitorder="little")[:n]
    assert bits.tolist() == want.tolist()
    assert int(d_count.item()) == int(want.sum())
    times = ctx.last_stage_times()
    assert set(times) >= {"trace_merkle", "pair_merkle", "transcript"} and all(v >= 0 for v in times.values())
    ctx.close()


def test_blob_produced_by_torch_kernels_right_before_verify(rsv):
    """the context runs on private non-blocking streams.  The batch is assembled by torch kernels
    on torch's current stream (repeat, a long chain of in-place XORs that cancel out, a scatter that tampers) and the
    verdict buffers are zeroed by torch, all WITHOUT a host synchronisation before the call: Context methods order
    their stream after torch's (rsv_ctx_wait_stream), so the verdicts must be those of the finished blob."""
    import torch
    proof = read_proof("recursive_proof_16_15.bin")
    n = 4096
    dev = torch.device("cuda:0")
    one = torch.from_numpy(np.frombuffer(proof, dtype=np.uint8).copy()).to(dev)
    offsets = torch.arange(n + 1, dtype=torch.int64, device=dev) * len(proof)
    ctx = rsv.Context(0)
    cfg = ctx.prepare_cfg(fixture_cfg("recursive_proof_16_15.bin"), n)
    tam = np.array([i for i in range(n) if i % 5 == 1], np.int64)
    pos = torch.from_numpy(tam * len(proof) + 60 + (tam * 7919) % (len(proof) - 68)).to(dev)
    torch.cuda.synchronize()
    for side in (False, True):  # torch's default stream, then a side stream made current
        stream = torch.cuda.Stream(dev) if side else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream):
            d_blob = one.repeat(n)                      # 446 MB written by a torch kernel
            for k in range(24):                         # ~20 GB of read-modify-write traffic in front of the verifier
                d_blob ^= (k % 7) + 1
            for k in range(24):
                d_blob ^= (k % 7) + 1
            d_blob[pos] = d_blob[pos] ^ 1
            d_acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
            d_reason = torch.full((n,), 7, dtype=torch.uint8, device=dev)
            ctx.verify_batch(d_blob, offsets, n, d_acc, d_reason, cfg=cfg)
            ctx.release_to_torch()                      # torch's stream now waits for the verdicts: no host block
            acc = d_acc.to("cpu", non_blocking=False).numpy()
        want = np.ones(n, np.uint8)
        want[tam] = 0
        assert np.array_equal(acc, want), (side, int((acc != want).sum()))
        del d_blob
    ctx.close()


def test_per_proof_cfg_index_outlives_the_python_call(rsv):
    """Context.verify_batch(cfg=[one PcsConfig per proof]) uploads the per-proof configuration index
    into a torch tensor that only the call's local PreparedCfg owns, and the parser reads it later, on the context's
    own stream.  Without a host synchronisation the tensor dies when the call returns; torch's caching allocator may
    then hand the block to the very next allocation on torch's stream, which is not ordered behind the context's.  The
    binding marks the tensor as in use by the context's stream (record_stream) and holds the last few.  Here: a long
    queue of torch work in front of the verifier (so that the parser runs late), the call, and immediately afterwards
    same-sized allocations filled with 0xFF (an index beyond the table: RSV_R_PARSE for every proof if it were read)."""
    import torch
    dev = torch.device("cuda:0")
    names = ["recursive_proof_16_15.bin", "level8-1.bin", "level12-1.bin", "level2-1.bin"]
    n = 2048
    batch = [read_proof(names[i % 4]) for i in range(n)]
    cfgs = [fixture_cfg(names[i % 4]) for i in range(n)]
    blob, offsets = rsv.pack(batch)
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    junk = torch.zeros(64 << 20, dtype=torch.int32, device=dev)
    ctx = rsv.Context(0)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for rep in range(4):
        for k in range(40):
            junk += k  # ~20 GB of traffic on torch's stream: the verifier's first kernel starts milliseconds later
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=cfgs)  # per-proof list, uploaded inside the call
        grabbed = [torch.full((n,), 0xFF, dtype=torch.uint8, device=dev) for _ in range(8)]  # would reuse a freed block
        ctx.synchronize()
        assert d_acc.cpu().numpy().tolist() == [1] * n, rep
        assert d_reason.cpu().numpy().tolist() == [0] * n, rep
        del grabbed
    ctx.close()


def _corrupt(proof: bytes, rng, region_end: int, n_bytes: int) -> bytes:
    b = bytearray(proof)
    for _ in range(n_bytes):
        b[int(rng.integers(0, region_end))] = int(rng.integers(0, 256))
    return bytes(b)


def test_fuzzed_headers_and_prefixes_match_oracle(rsv):
    """Robustness: random byte corruption of the header / length-prefix areas (and of whole proofs) must never
    fault and must give the oracle's verdict and reason (mostly RSV_R_PARSE)."""
    rng = np.random.default_rng(11)
    small = read_proof("small_proof.bin")
    big = read_proof("level2-1.bin")
    batch = []
    for k in range(48):
        batch.append(_corrupt(small, rng, 64, 1 + k % 3))           # header words
    for k in range(48):
        batch.append(_corrupt(small, rng, 3620, 1 + k % 4))         # constant-shape prefix + first prefixes
    for k in range(32):
        batch.append(_corrupt(small, rng, len(small), 8))           # anywhere
    # length prefixes of the variable part: overwrite a u64 prefix with a huge / odd count
    for off in (3580, 3588, 3596):
        for val in (0, 1, 5, 0xFFFFFFFF, 1 << 40):
            b = bytearray(small)
            b[off:off + 8] = int(val).to_bytes(8, "little")
            batch.append(bytes(b))
    acc, reason = rsv.verify_batch(batch, fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    oacc, oreason = ob.verify_batch(batch, fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc.tolist() == oacc.tolist()
    assert reason.tolist() == oreason.tolist()
    batch2 = [_corrupt(big, rng, 3620, 2) for _ in range(16)] + [big]
    acc, reason = rsv.verify_batch(batch2, fixture_cfg("level2-1.bin"))
    oacc, oreason = ob.verify_batch(batch2, fixture_cfg("level2-1.bin"))
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    assert acc[-1] == 1


def test_chunked_workspace_matches_unchunked(rsv):
    """A small workspace budget forces the per-query stages to run in several chunks."""
    import torch
    proof = read_proof("small_proof.bin")
    n = 2600
    batch = [ob.tamper(proof, i) if i % 11 == 4 else proof for i in range(n)]
    blob, offsets = rsv.pack(batch)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    results = []
    for budget in (64, 8192):
        ctx = rsv.Context(0)
        ctx.set_option("ws_budget_mb", budget)
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=fixture_cfg("small_proof.bin"), inputs=[(1, (1, 0, 0, 0))])
        ctx.synchronize()
        results.append((d_acc.cpu().numpy().tolist(), d_reason.cpu().numpy().tolist()))
        ctx.close()
    assert results[0] == results[1]
    want = [0 if i % 11 == 4 else 1 for i in range(n)]
    assert results[0][0] == want


def test_cap_disabled_matches_cap_enabled(rsv, knobs):
    """tree_cap = off walks every path to the root; the dense top-of-tree cap must give identical verdicts."""
    proof = read_proof("recursive_proof_16_15.bin")
    # several proofs per bucket, buckets with fewer FRI layers than the launch's deepest one (their slots' cap nodes must
    # not be touched by the grid rows of layers they do not have — a violation shows only at scale, when such rows run after
    # the rightful ones: the 53 248-proof chain of tests/test_full_size_parity.py caught it), tampered copies of every shape
    extra = ["level12-1.bin", "level1-5.bin", "level12-1.bin", "level9-1.bin", "level12-1.bin", "level2-1.bin", "level9-1.bin"]
    batch = [ob.tamper(proof, i) for i in range(48)] + [ob.tamper(read_proof(f), 3 + k) for k, f in enumerate(extra)] + \
            [read_proof(f) for f in extra] + [proof, read_proof("level1-5.bin"), read_proof("level12-1.bin")]
    cfgs = [fixture_cfg("recursive_proof_16_15.bin")] * 48 + [fixture_cfg(f) for f in extra] * 2 + \
           [fixture_cfg("recursive_proof_16_15.bin"), fixture_cfg("level1-5.bin"), fixture_cfg("level12-1.bin")]
    knobs.set("tree_cap", "off")
    a0, r0 = rsv.verify_batch(batch, cfgs)
    knobs.set("tree_cap", "on")
    knobs.set("cap_top", "off")
    a1, r1 = rsv.verify_batch(batch, cfgs)
    assert a0.tolist() == a1.tolist() and r0.tolist() == r1.tolist()
    assert a1[-3:].tolist() == [1, 1, 1]
    # the last levels of every tree in k_cap_top (what batches of >= 8 192 proofs run): Lt = 2 for the 16- and 8-query
    # proofs, 3 for the 80-query one; the oracle's verdicts for the tampered copies
    knobs.set("cap_top", "on")
    a2, r2 = rsv.verify_batch(batch, cfgs)
    assert a0.tolist() == a2.tolist() and r0.tolist() == r2.tolist()
    wa, wr = ob.verify_batch(batch, cfgs)
    assert a2.tolist() == wa.tolist() and r2.tolist() == wr.tolist()


@pytest.mark.parametrize("pace", ["paced", "unpaced", "row16"])
@pytest.mark.parametrize("order", ["on", "off"])
def test_tree_kernel_forms_match_oracle(rsv, knobs, pace, order):
    """The Merkle kernels' three forms (lane form with / without wait states, row form on virtual lanes of 16 threads;
    picked by batch size in production) and the two assignments of FRI trees to grid rows (dealt out by depth for a launch that is resident all at once / row y =
    tree y), forced on one batch of several shapes with tampered copies: verdicts and reasons == the oracle's."""
    names = ["recursive_proof_16_15.bin", "level1-5.bin", "level12-1.bin", "level9-1.bin", "level3-1.bin"]
    batch, cfgs = [], []
    for k in range(70):
        nm = names[k % len(names)] if k % 7 == 0 else names[0]
        pr = read_proof(nm)
        batch.append(ob.tamper(pr, k) if k % 3 == 1 else pr)
        cfgs.append(fixture_cfg(nm))
    knobs.set("tree_pace", pace)
    knobs.set("pair_order", order)
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist() and 20 < int(acc.sum()) < 60
    # one configuration (the device-side slot order, where the tree order is computed from the configuration alone)
    one = [b for b, c in zip(batch, cfgs) if c.n_queries == 16 and c.log_blowup_factor == 5][:40]
    acc1, reason1 = rsv.verify_batch(one, fixture_cfg(names[0]))
    o1, r1 = ob.verify_batch(one, fixture_cfg(names[0]))
    assert acc1.tolist() == o1.tolist() and reason1.tolist() == r1.tolist()
    # several configurations of at most 32 queries each (the row form's limit: the first batch above, which holds an
    # 80-query shape, falls back to the unpaced lane form under "row16"), tampered copies among them
    few = ["level2-1.bin", "level8-1.bin", "level12-1.bin", "level10-1.bin", "small_proof.bin", "level6-1.bin"]
    batch2 = [read_proof(few[k % len(few)]) for k in range(30)]
    batch2 = [ob.tamper(pr, 3 * k) if k % 4 == 2 else pr for k, pr in enumerate(batch2)]
    cfgs2 = [fixture_cfg(few[k % len(few)]) for k in range(30)]
    acc2, reason2 = rsv.verify_batch(batch2, cfgs2)
    o2, r2 = ob.verify_batch(batch2, cfgs2)
    assert acc2.tolist() == o2.tolist() and reason2.tolist() == r2.tolist() and 10 < int(acc2.sum()) < 30
    # and with the last levels of the cap in k_cap_top (production: batches of >= 1 024 proofs), fed by this form's cap
    knobs.set("cap_top", "on")
    acc3, reason3 = rsv.verify_batch(one, fixture_cfg(names[0]))
    assert acc3.tolist() == o1.tolist() and reason3.tolist() == r1.tolist()


@pytest.mark.parametrize("mid", ["auto", "on", "off"])
@pytest.mark.parametrize("pace", ["paced", "unpaced"])
def test_cap_mid_forms_match_oracle(rsv, knobs, mid, pace):
    """The top of every Merkle tree with the cap kernels (production: batches of >= 1 024 proofs): a bucket hands its nodes over
    at the cap level and k_cap_mid (a lane per subtree) + k_cap_top finish the tree — for the query counts whose dense cap
    levels fill the tree kernels' waves badly (auto: 80, 27, 11, 10), for every bucket (on), for none (off).  Every shape of
    the recursion chain with tampered copies, several proofs per workgroup and a ragged last one, then one configuration
    per call (the device-side slot order): verdicts and reasons == the oracle's."""
    names = ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin", "level7-1.bin",
             "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin", "level13-1.bin"]
    knobs.set("tree_pace", pace)
    knobs.set("cap_top", "on")
    knobs.set("cap_mid", mid)
    batch, cfgs = [], []
    for k in range(13 * 7 + 5):
        nm = names[k % 13]
        pr = read_proof(nm)
        batch.append(ob.tamper(pr, k) if k % 4 == 1 else pr)
        cfgs.append(fixture_cfg(nm))
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist() and 40 < int(acc.sum()) < 90
    # every proof of 16 and fewer queries tampered, the others genuine
    batch2 = [ob.tamper(pr, 7 * k + 3) if c.n_queries <= 16 else read_proof(names[k % 13]) for k, (pr, c) in enumerate(zip(batch, cfgs))]
    acc2, reason2 = rsv.verify_batch(batch2, cfgs)
    o2, r2 = ob.verify_batch(batch2, cfgs)
    assert acc2.tolist() == o2.tolist() and reason2.tolist() == r2.tolist()
    assert all(a == (c.n_queries > 16) for a, c in zip(acc2.tolist(), cfgs))
    for nm, n in (("level1-5.bin", 7), ("level2-1.bin", 19), ("level10-1.bin", 53), ("recursive_proof_16_15.bin", 35), ("level13-1.bin", 70)):
        pr = read_proof(nm)
        one = [ob.tamper(pr, 5 * k) if k % 3 == 2 else pr for k in range(n)]
        a1, r1 = rsv.verify_batch(one, fixture_cfg(nm))
        o1, q1 = ob.verify_batch(one, fixture_cfg(nm))
        assert a1.tolist() == o1.tolist() and r1.tolist() == q1.tolist() and int(a1.sum()) >= n // 2, nm


@pytest.mark.parametrize("trees", ["paced", "row16"])
@pytest.mark.parametrize("name", ["small_proof.bin", "recursive_proof_16_15.bin", "level2-1.bin", "level13-1.bin"])
def test_trace_paths_match_oracle(rsv, manifest, knobs, name, trees):
    """per-query authentication paths (transcript order) emitted by the GPU == oracle's; from the lane form and
    from the row form of the tree kernels (what a batch of two proofs takes by itself)."""
    knobs.set("tree_pace", trees)
    entry = next(e for e in manifest if e["file"] == name)
    proof = read_proof(name)
    nq = entry["n_queries"]
    M = max(entry["log_size_plonk"] + 1, entry["log_size_poseidon"] + 2) + entry["log_blowup_factor"]
    inputs = entry_inputs(entry)
    osib, opos, depth = ob.trace_paths(proof, nq, M, inputs)
    sib, pos, acc, reason = rsv.trace_paths([proof, proof], fixture_cfg(name), nq, M, inputs)
    assert acc.tolist() == [1, 1] and reason.tolist() == [0, 0]
    for k in range(2):
        assert np.array_equal(pos[k], opos)
        for t in range(4):
            d = int(depth[t])
            assert np.array_equal(sib[k, t, :, :d, :], osib[t, :, :d, :]), (k, t)
    # shape mismatch is an API error, not a verdict
    with pytest.raises(rsv.RsvError) as e:
        rsv.trace_paths([proof], fixture_cfg(name), nq, M + 1, inputs)
    assert e.value.code == -2


@pytest.mark.parametrize("mode", ["row", "lane"])
def test_transcript_kernels_row_and_lane(rsv, manifest, knobs, mode):
    """Both transcript kernels (one proof per 16-lane DPP row / one proof per lane) against the oracle."""
    knobs.set("transcript_form", mode)
    for entry in manifest:
        proof = read_proof(entry["file"])
        assert rsv.transcript(proof) == rsv._parse_transcript(ob.transcript_raw(proof)), (mode, entry["file"])
    proof = read_proof("recursive_proof_16_15.bin")
    batch = [ob.tamper(proof, i) for i in range(40)] + [proof, read_proof("level1-5.bin"), read_proof("level13-1.bin")]
    cfgs = [fixture_cfg("recursive_proof_16_15.bin")] * 41 + [fixture_cfg("level1-5.bin"), fixture_cfg("level13-1.bin")]
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()


def test_canonicity_is_checked_where_words_are_read(rsv):
    """Every field-element word of a proof must be < P, else RSV_R_PARSE.  Rounds 1-2 read every proof a second time
    for this (a scan kernel); now each stage checks the words it reads anyway and only proofs whose witness lists have
    the wrong length are re-read (csrc/layout.hpp).  Non-canonical words at the start, in the middle and at the end of
    every section and in the exempt words (nonce halves, final word); then proofs that are wrong twice — a list of the
    wrong length AND a non-canonical word where no stage reads — whose reason must still be PARSE, as the oracle says."""
    proof = read_proof("recursive_proof_16_15.bin")
    cfg = fixture_cfg("recursive_proof_16_15.bin")
    words = np.frombuffer(proof, np.uint32)
    lay = ob.proof_layout(proof)
    spots = sorted({2, 9, 17, 48, 60, 894, len(words) - 2, len(words) - 1, lay["nonce_word"], lay["nonce_word"] + 1, lay["nonce_word"] - 1,
                    lay["nonce_word"] + 2} | {pos + 2 for pos, n, _ in lay["prefixes"] if n and pos + 2 < len(words)}
                   | {pos + 2 + k for pos, n, what in lay["prefixes"] if n > 4 and "inner_layers" not in what and what != "decommitments"
                      and what != "queried_values" for k in (n // 2, n - 1)})
    batch = [proof]
    for k in range(3 * len(spots)):
        w = words.copy()
        w[spots[k % len(spots)]] = [0x7FFFFFFF, 0x80000000, 0xFFFFFFFF][k // len(spots)]
        batch.append(w.tobytes())
    n_spot = len(batch)
    cfgs = [cfg] * n_spot
    for name in ("recursive_proof_16_15.bin", "level2-1.bin", "level12-1.bin"):
        mut = [b for _, b in ob.noncanonical_structural_mutants(read_proof(name))]
        batch += mut
        cfgs += [fixture_cfg(name)] * len(mut)
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    assert acc[0] == 1 and set(oreason[n_spot:].tolist()) == {1}
    # the same mutants one proof per call (another grid, another workgroup geometry)
    for b, c_, oa, orr in list(zip(batch, cfgs, oacc, oreason))[n_spot::7]:
        a1, r1 = rsv.verify_batch([b], c_)
        assert (int(a1[0]), int(r1[0])) == (int(oa), int(orr))


@pytest.mark.parametrize("form", ["row", "lane"])
@pytest.mark.parametrize("split", ["whole", "split"])
def test_transcript_in_one_piece_and_split(rsv, knobs, split, form):
    """Either transcript form as one launch and as front (next to the parser, before any section offset is known) +
    back: the same verdicts, reasons and transcript rows, also for buffers the front half has to leave alone
    (empty, truncated inside the fixed-offset part, cut right behind it, misaligned length), and a non-canonical word
    in the front's part of the proof, whose finding travels to the back half."""
    knobs.set("transcript_form", form)
    knobs.set("transcript_split", split)
    proof = read_proof("recursive_proof_16_15.bin")
    cfg = fixture_cfg("recursive_proof_16_15.bin")
    batch = [proof, ob.tamper(proof, 3), b"", proof[:64], proof[:3576], proof[:3584], proof[:3620], proof[:len(proof) - 4],
             proof[:50000], proof, read_proof("level6-1.bin")]
    cfgs = [cfg] * 10 + [fixture_cfg("level6-1.bin")]
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() == [1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1]
    assert reason.tolist() == oreason.tolist()
    rows = rsv.transcript_batch(batch, cfgs)
    for k in (0, 9, 10):
        assert np.array_equal(rows[k], _row_from_raw(ob.transcript_raw(batch[k])))
    words = np.frombuffer(proof, np.uint32)
    over = []
    for spot in (3, 20, 56, 894):  # a claimed sum, a commitment word, sampled values: all absorbed by the front half
        w = words.copy()
        w[spot] = 0x80000001
        over.append(w.tobytes())
    acc, reason = rsv.verify_batch(over + [proof], cfg)
    oacc, oreason = ob.verify_batch(over + [proof], cfg)
    assert acc.tolist() == oacc.tolist() == [0, 0, 0, 0, 1] and reason.tolist() == oreason.tolist() == [1, 1, 1, 1, 0]


@pytest.mark.parametrize("trees", ["paced", "row16"])
@pytest.mark.parametrize("name", ["small_proof.bin", "recursive_proof_16_15.bin", "level7-1.bin", "level2-1.bin", "level13-1.bin"])
def test_fri_paths_match_oracle(rsv, manifest, knobs, name, trees):
    """per-query pair paths of every FRI tree (transcript order) emitted by the GPU == oracle's; lane form and
    row form of the tree kernels."""
    knobs.set("tree_pace", trees)
    entry = next(e for e in manifest if e["file"] == name)
    proof = read_proof(name)
    lay = ob.proof_layout(proof)
    nq, n_inner = entry["n_queries"], lay["n_inner"]
    M = max(entry["log_size_plonk"] + 1, entry["log_size_poseidon"] + 2) + entry["log_blowup_factor"]
    inputs = entry_inputs(entry)
    osib, ocols = ob.fri_paths(proof, nq, M, 1 + n_inner, inputs)
    sib, cols, acc, reason = rsv.fri_paths([proof, proof], fixture_cfg(name), nq, M, n_inner, inputs)
    assert acc.tolist() == [1, 1] and reason.tolist() == [0, 0]
    for k in range(2):
        assert np.array_equal(cols[k], ocols)
        for s2 in range(1 + n_inner):
            d = M if s2 == 0 else M - s2
            assert np.array_equal(sib[k, s2, :, :d - 1, :], osib[s2, :, :d - 1, :]), (k, s2)


def _prefix_mutants(proof):
    out = []
    for pos, n, _ in ob.proof_layout(proof)["prefixes"]:
        for val in {max(n - 1, 0), n + 1, 0, 0xFFFFFFFF, (1 << 32) + n, 8 * n + 3} - {n}:
            b = bytearray(proof)
            b[4 * pos:4 * pos + 8] = int(val).to_bytes(8, "little")
            out.append(bytes(b))
    return out


@pytest.mark.parametrize("name", ["small_proof.bin", "level12-1.bin", "level2-1.bin"])
def test_every_length_prefix_mutated_matches_oracle(rsv, manifest, name):
    """Robustness: every u64 length prefix of the variable part (decommitments, queried values, FRI layers, last
    layer) set to n-1, n+1, 0, 2^32-1, 2^32+n and 8n+3.  No launch may fault; verdict and reason == oracle's."""
    entry = next(e for e in manifest if e["file"] == name)
    inputs = entry_inputs(entry)
    proof = read_proof(name)
    batch = _prefix_mutants(proof) + [proof]
    acc, reason = rsv.verify_batch(batch, fixture_cfg(name), inputs)
    oacc, oreason = ob.verify_batch(batch, fixture_cfg(name), inputs)
    assert acc.tolist() == oacc.tolist()
    assert reason.tolist() == oreason.tolist()
    assert acc[-1] == 1 and int(acc[:-1].sum()) == 0


@pytest.mark.parametrize("name", ["small_proof.bin", "level12-1.bin", "level2-1.bin", "level1-5.bin"])
def test_structural_mutants_match_oracle(rsv, manifest, name):
    """Well-formed proofs whose witness lists are one element short / long, emptied, rotated or moved between trees
    and layers (tests/oracle_binding.py::structural_mutants): they pass the parser and must fail in the stage that
    consumes the list, with the oracle's reason, and without any out-of-range access."""
    entry = next(e for e in manifest if e["file"] == name)
    inputs = entry_inputs(entry)
    proof = read_proof(name)
    mut = ob.structural_mutants(proof)
    batch = [b for _, b in mut] + [proof]
    acc, reason = rsv.verify_batch(batch, fixture_cfg(name), inputs)
    oacc, oreason = ob.verify_batch(batch, fixture_cfg(name), inputs)
    bad = [(mut[i][0], int(reason[i]), int(oreason[i])) for i in range(len(mut)) if reason[i] != oreason[i]]
    assert not bad, bad
    assert acc.tolist() == oacc.tolist()
    assert acc[-1] == 1 and int(acc[:-1].sum()) == 0


def _row_from_raw(raw):
    """rsv_transcript's packed layout (oracle: rsvo_transcript) -> the fixed-stride row of rsv_transcript_batch."""
    row = np.zeros(284, np.uint32)
    if raw[0] == 1:  # RSV_R_PARSE
        row[0] = 1
        return row
    na, nq = int(raw[1]), int(raw[2])
    row[:40] = raw[:40]
    row[40:40 + 4 * na] = raw[40:40 + 4 * na]
    row[156:156 + nq] = raw[40 + 4 * na:40 + 4 * na + nq]
    return row


def test_transcript_batch_matches_oracle(rsv, manifest):
    """FiatShamirHints rows for a mixed-shape batch: all fixtures, the SHA-256-channel fixture (parse reject), a
    proof with a flipped commitment byte (PoW reject) and garbage."""
    proofs = [read_proof(e["file"]) for e in manifest]
    cfgs = [fixture_cfg(e["file"]) for e in manifest]
    bad = bytearray(proofs[0]); bad[70] ^= 1
    proofs += [read_proof("hybrid_hash.bin"), bytes(bad), b"\x00" * 64]
    cfgs += [fixture_cfg("hybrid_hash.bin"), cfgs[0], cfgs[0]]
    rows = rsv.transcript_batch(proofs, cfgs)
    for i, pr in enumerate(proofs):
        want = _row_from_raw(ob.transcript_raw(pr)) if len(pr) >= 4096 else _row_from_raw(np.array([1], np.uint32))
        assert np.array_equal(rows[i], want), i
    assert rsv.transcript_batch([b"\x00" * 64, b"\x01" * 8], cfgs[0])[:, 0].tolist() == [1, 1]  # nothing parses
    # a proof presented under another fixture's configuration: its row says PARSE, nothing else
    wrong = rsv.transcript_batch([proofs[0], proofs[1]], [cfgs[1], cfgs[1]])
    assert wrong[0, 0] == 1 and not wrong[0, 1:].any() and np.array_equal(wrong[1], rows[1])


def test_verify_hints_one_pass(rsv, manifest):
    """rsv_verify_hints_dev: verdicts + transcript rows + trace paths + FRI paths from ONE pass == the three
    single-purpose entry points."""
    import torch
    entry = next(e for e in manifest if e["file"] == "recursive_proof_16_15.bin")
    proof = read_proof(entry["file"])
    lay = ob.proof_layout(proof)
    nq, n_inner = entry["n_queries"], lay["n_inner"]
    M = max(entry["log_size_plonk"] + 1, entry["log_size_poseidon"] + 2) + entry["log_blowup_factor"]
    n = 40
    batch = [ob.tamper(proof, i) if i % 9 == 4 else proof for i in range(n)]
    blob, offsets = rsv.pack(batch)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_tr = torch.zeros((n, rsv.TRANSCRIPT_WORDS), dtype=torch.int32, device=dev)
    d_ts = torch.zeros((n, 4, nq, M, 8), dtype=torch.int32, device=dev)
    d_tp = torch.zeros((n, 4, nq), dtype=torch.int32, device=dev)
    d_tc = torch.zeros((n, 4, nq, 64), dtype=torch.int32, device=dev)
    d_fs = torch.zeros((n, 1 + n_inner, nq, M, 8), dtype=torch.int32, device=dev)
    d_fc = torch.zeros((n, 1 + n_inner, nq, 3, 8), dtype=torch.int32, device=dev)
    d_ff = torch.zeros((n, 3, nq, 4), dtype=torch.int32, device=dev)
    ctx = rsv.Context(0)
    cfg = fixture_cfg(entry["file"])
    ctx.verify_hints(d_blob, d_off, n, d_acc, d_reason, cfg=cfg, shape=(nq, M, n_inner), d_transcript=d_tr, d_trace_sib=d_ts,
                     d_trace_pos=d_tp, d_trace_cols=d_tc, d_fri_sib=d_fs, d_fri_cols=d_fc, d_fri_folded=d_ff)
    ctx.synchronize()
    oacc, oreason = ob.verify_batch(batch, cfg)
    assert d_acc.cpu().numpy().tolist() == oacc.tolist() and d_reason.cpu().numpy().tolist() == oreason.tolist()
    tr = d_tr.cpu().numpy().view(np.uint32)
    assert np.array_equal(tr, rsv.transcript_batch(batch, cfg))
    tsib, tpos, _, _ = rsv.trace_paths(batch, cfg, nq, M)
    fsib, fcols, _, _ = rsv.fri_paths(batch, cfg, nq, M, n_inner)
    ok = np.nonzero(oacc)[0]
    assert np.array_equal(d_ts.cpu().numpy().view(np.uint32)[ok], tsib[ok]) and np.array_equal(d_tp.cpu().numpy().view(np.uint32)[ok], tpos[ok])
    assert np.array_equal(d_fs.cpu().numpy().view(np.uint32)[ok], fsib[ok]) and np.array_equal(d_fc.cpu().numpy().view(np.uint32)[ok], fcols[ok])
    assert np.array_equal(d_ff.cpu().numpy().view(np.uint32)[ok[0]], ob.fri_folded(proof))
    # SinglePathMerkleProof::columns, and the whole struct through the path verifier: every path -> its commitment
    tcols = d_tc.cpu().numpy().view(np.uint32)
    assert np.array_equal(tcols[ok[0]], ob.trace_cols(proof))
    words = np.frombuffer(proof, dtype=np.uint32)
    A, B = entry["log_size_plonk"] + entry["log_blowup_factor"], entry["log_size_poseidon"] + entry["log_blowup_factor"]
    for t in range(4):
        d = M if t == 3 else max(A, B)
        n_cols_at = [0] * (d + 1)
        if t == 3:
            n_cols_at[M] = 8
        else:
            n_cols_at[A] += [10, 12, 8][t]
            n_cols_at[B] += [40, 48, 8][t]
        roots = rsv.merkle_path_root(tpos[ok[1], t], tsib[ok[1], t][:, :d, :], tcols[ok[1], t][:, :sum(n_cols_at)], n_cols_at)
        assert all(r.tolist() == words[17 + 8 * t:25 + 8 * t].tolist() for r in roots), t
    # a shape that does not match the batch is an API error, not a verdict
    with pytest.raises(rsv.RsvError):
        ctx.verify_hints(d_blob, d_off, n, d_acc, d_reason, cfg=cfg, shape=(nq + 1, M, n_inner), d_trace_sib=d_ts, d_trace_pos=d_tp)
    ctx.close()


@pytest.mark.parametrize("chunk_mb", [1, 256])
def test_verify_batch_host_pipeline(rsv, manifest, chunk_mb):
    """rsv_verify_batch_host: one host buffer per proof, gather -> upload -> verify pipelined over chunks.  A 1 MB
    chunk size forces dozens of chunks through the three-slot ring; verdicts == oracle's, in input order."""
    proofs, cfgs = [], []
    for e in manifest:
        pr = read_proof(e["file"])
        if entry_inputs(e) == list(rsv.STANDARD_INPUTS):
            proofs += [pr, ob.tamper(pr, len(proofs)), pr]
            cfgs += [fixture_cfg(e["file"])] * 3
    proofs += [read_proof("hybrid_hash.bin"), b"\x00" * 64, b""]
    cfgs += [fixture_cfg("hybrid_hash.bin"), cfgs[0], cfgs[0]]
    ctx = rsv.Context(0)
    ctx.set_option("host_chunk_mb", chunk_mb)
    acc, reason = ctx.verify_batch_host(proofs, cfgs)
    oacc, oreason = ob.verify_batch(proofs, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    assert int(acc.sum()) > 10
    # a length that is not a whole number of words is a malformed proof like any other: RSV_R_PARSE for that proof only
    acc1, reason1 = ctx.verify_batch_host([proofs[0], b"\x00" * 6, proofs[0]], cfgs[0])
    assert acc1.tolist() == [1, 0, 1] and reason1.tolist() == [0, 1, 0]
    # an absurdly long buffer is not uploaded: RSV_R_PARSE, and its neighbours are unaffected
    big = np.zeros(40 << 20, np.uint8)
    acc2, reason2 = ctx.verify_batch_host([proofs[0], big, proofs[0]], cfgs[0])
    assert acc2.tolist() == [1, 0, 1] and reason2.tolist() == [0, 1, 0]
    assert ctx.verify_batch_host([], cfgs[0])[0].size == 0
    with pytest.raises(TypeError):
        ctx.verify_batch_host(proofs, None)
    ctx.close()


@pytest.mark.parametrize("chunk_mb", [1, 256])
def test_verify_batch_host_from_a_pinned_arena(rsv, manifest, chunk_mb):
    """rsv_host_alloc: proofs read back to back into the library's pinned arena are uploaded from where they are (no
    gather copy); chunks that do not qualify — a proof outside the arena, a gap, a buffer that is not a whole number of
    words — take the gather path, the two mixed chunk by chunk in one call.  Verdicts and reasons == the oracle's either
    way, and == the plain path's."""
    proofs, cfgs = [], []
    for e in manifest:
        pr = read_proof(e["file"])
        if entry_inputs(e) == list(rsv.STANDARD_INPUTS):
            proofs += [pr, ob.tamper(pr, len(proofs)), pr]
            cfgs += [fixture_cfg(e["file"])] * 3
    arena = rsv.HostArena(sum(len(p) for p in proofs) + 4096)
    hb = arena.pack(proofs)                      # every proof inside the arena, contiguous in job order
    ctx = rsv.Context(0)
    ctx.set_option("host_chunk_mb", chunk_mb)
    oacc, oreason = ob.verify_batch(proofs, cfgs)
    acc, reason = ctx.verify_batch_host(hb, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist() and int(acc.sum()) > 10
    # the same job with every fifth proof held OUTSIDE the arena (pageable memory) and one odd-length buffer inside it
    mixed = list(hb.keep)
    for i in range(2, len(mixed), 5):
        mixed[i] = np.frombuffer(proofs[i], dtype=np.uint8).copy()
    acc2, reason2 = ctx.verify_batch_host(mixed, cfgs)
    assert acc2.tolist() == oacc.tolist() and reason2.tolist() == oreason.tolist()
    odd = list(hb.keep)
    odd[1] = odd[1][:len(odd[1]) - 2]           # not a whole number of words: RSV_R_PARSE, neighbours unaffected
    acc3, reason3 = ctx.verify_batch_host(odd, cfgs)
    want3, wantr3 = oacc.copy(), oreason.copy()
    want3[1], wantr3[1] = 0, 1
    assert acc3.tolist() == want3.tolist() and reason3.tolist() == wantr3.tolist()
    # proofs in the arena but NOT in job order (reversed): contiguity fails, the gather path takes them
    rev = list(reversed(hb.keep))
    acc4, reason4 = ctx.verify_batch_host(rev, list(reversed(cfgs)))
    assert acc4.tolist() == oacc[::-1].tolist() and reason4.tolist() == oreason[::-1].tolist()
    # through the multi-context entry point as well (three contexts on the one device)
    mc = rsv.MultiContext([0, 0, 0])
    acc5, reason5, _, count5 = mc.verify_batch_host(hb, cfgs)
    assert acc5.tolist() == oacc.tolist() and reason5.tolist() == oreason.tolist() and count5 == int(oacc.sum())
    mc.close()
    ctx.close()
    arena.close()
    assert rsv.lib.rsv_host_alloc(0, None) == -1 and rsv.lib.rsv_host_free(None) is None


def test_mutant_corpus_matches_oracle(rsv, manifest):
    """Parity fuzz: structural mutants, every length prefix perturbed, random byte corruption and truncations of
    six fixtures of different shapes (n_queries 8..80) in ONE mixed batch: verdict and reason == oracle's."""
    from tests.mutants import mutants_of
    rng = np.random.default_rng(5)
    batch, cfgs = [], []
    for name in ("recursive_proof_16_15.bin", "level1-5.bin", "level5-1.bin", "level8-1.bin", "level10-1.bin", "level13-1.bin"):
        proof = read_proof(name)
        mut = mutants_of(proof, rng, 60) + [proof]
        batch += mut
        cfgs += [fixture_cfg(name)] * len(mut)
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    diff = np.nonzero((acc != oacc) | (reason != oreason))[0]
    assert diff.size == 0, [(int(i), int(reason[i]), int(oreason[i])) for i in diff[:10]]
    assert int(acc.sum()) >= 6


def test_mutants_behind_the_proof_of_work_match_oracle(rsv):
    """With the real pow_bits every mutant of a transcript-absorbed section stops at the proof of work.  Here the
    fixtures carry pow_bits = 0 in their header and are verified under pow_bits = 0 (everything else as in the reference's
    literals), so corrupted commitments, sampled values, FRI commitments and last-layer coefficients reach the logup
    and composition checks and — with query positions that no longer match the decommitments — the plan, Merkle and
    FRI kernels.  Verdict and reason == oracle's; reasons 3 and 4 must actually occur."""
    from tests.conftest import Cfg
    rng = np.random.default_rng(17)
    batch, cfgs = [], []
    for name in ("recursive_proof_16_15.bin", "level1-5.bin", "level6-1.bin", "level12-1.bin"):
        w = np.frombuffer(read_proof(name), np.uint32).copy()
        w[10] = 0  # W_POW_BITS
        proof = w.tobytes()
        c = fixture_cfg(name)
        cfg0 = Cfg(0, c.log_blowup_factor, c.log_last_layer_degree_bound, c.n_queries)
        n_head = 4 * ob.proof_layout(proof)["nonce_word"]  # everything in front of the nonce is absorbed by the transcript
        mut = [proof]
        for k in range(220):
            b = bytearray(proof)
            for _ in range(1 + k % 3):
                # k % 4 == 1: the first two commitments (mixed before z and alpha are drawn -> logup); k odd: the
                # transcript-absorbed front part; k even: anywhere
                pos = int(rng.integers(68, 132)) if k % 4 == 1 else int(rng.integers(68, n_head)) if k % 2 else int(rng.integers(0, len(b)))
                b[pos] = int(rng.integers(0, 256))
            mut.append(bytes(b))
        batch += mut
        cfgs += [cfg0] * len(mut)
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    diff = np.nonzero((acc != oacc) | (reason != oreason))[0]
    assert diff.size == 0, [(int(i), int(reason[i]), int(oreason[i])) for i in diff[:10]]
    counts = np.bincount(reason, minlength=13)
    assert int(acc.sum()) >= 4 and counts[2] == 0 and counts[3] > 0 and counts[4] > 0 and counts[6:12].sum() > 0


def test_field_ops_match_oracle(rsv):
    """Rows a1/a2 on their own: every RSV_F_* operation on 2^16 random elements plus edge values == oracle."""
    rng = np.random.default_rng(21)
    n = 1 << 16
    a = rng.integers(0, P, (n, 4), dtype=np.uint32)
    b = rng.integers(0, P, (n, 4), dtype=np.uint32)
    edge = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [P - 1] * 4, [0, 0, 1, 0], [0, 1, 0, 0], [P - 1, 0, 0, 0], [0, 0, 0, 1]], np.uint32)
    a[:7], b[:7] = edge, edge[::-1]
    for op in range(10):
        bb = b if op in (0, 1, 2, 4, 6) else None
        assert np.array_equal(rsv.field_op(op, a, bb), ob.field_op(op, a, bb)), op
    e = np.zeros((256, 4), np.uint32)
    e[:, 0] = rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)
    e[:4, 0] = [0, 1, 2, 0xFFFFFFFF]
    assert np.array_equal(rsv.field_op(rsv.F_QPOW, a[:256], e), ob.field_op(10, a[:256], e))
    with pytest.raises(rsv.RsvError):
        rsv.field_op(rsv.F_QMUL, [[P, 0, 0, 0]], [[1, 0, 0, 0]])
    assert rsv.field_op(rsv.F_QMUL, np.zeros((0, 4), np.uint32), np.zeros((0, 4), np.uint32)).shape == (0, 4)


@pytest.mark.parametrize("log_size", [1, 2, 5, 13, 21, 26, 30])
def test_domain_points_match_oracle(rsv, log_size):
    """Row a8 (reference test primitives/circle/src/lib.rs:264): bit-reversed circle-domain points."""
    rng = np.random.default_rng(log_size)
    q = rng.integers(0, 1 << log_size, 3000, dtype=np.uint64).astype(np.uint32)
    q[:4] = [0, 1, (1 << log_size) - 1, (1 << log_size) >> 1]
    got = rsv.domain_points(log_size, q)
    assert np.array_equal(got, ob.domain_points(log_size, q))
    x, y = got[:, 0].astype(object), got[:, 1].astype(object)
    assert all((int(a) * int(a) + int(b) * int(b)) % P == 1 for a, b in zip(x[:64], y[:64]))


@pytest.mark.parametrize("log_n", [0, 1, 3, 4, 5, 8, 12])
def test_line_eval_matches_oracle(rsv, log_n):
    """Last layer of a12 (reference test primitives/line/src/lib.rs:82): LinePoly evaluation."""
    rng = np.random.default_rng(100 + log_n)
    coeffs = rng.integers(0, P, (1 << log_n, 4), dtype=np.uint32)
    xs = rng.integers(0, P, 2000, dtype=np.uint32)
    xs[:3] = [0, 1, P - 1]
    assert np.array_equal(rsv.line_eval(coeffs, xs), ob.line_eval(coeffs, xs))


def test_plan_kernels_serial_and_parallel(rsv, manifest, knobs):
    """The decommitment plan has two implementations — one lane per (proof, query) with bitmask popcounts (default)
    and the one-lane-per-proof walk it replaced (plan_form = serial).  Both must give the oracle's verdicts on the
    whole fixture set plus structural mutants (whose rejection reasons depend on the plan's witness counts)."""
    std = [e for e in manifest if entry_inputs(e) == list(rsv.STANDARD_INPUTS)]
    proofs = [read_proof(e["file"]) for e in std]
    batch, cfgs = list(proofs), [fixture_cfg(e["file"]) for e in std]
    for e, pr in list(zip(std, proofs))[:4]:
        mut = [b for _, b in ob.structural_mutants(pr)]
        batch += mut
        cfgs += [fixture_cfg(e["file"])] * len(mut)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    for mode in ("serial", "parallel"):
        knobs.set("plan_form", mode)
        acc, reason = rsv.verify_batch(batch, cfgs)
        assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist(), mode


def test_reject_fixtures_behind_the_proof_of_work(rsv):
    """Rejection stages that random corruption never reaches (it stops at the proof of work): the OODS composition
    identity and the duplicate-query assertion, with the re-ground fixtures of tests/golden/make_reject_fixtures.py."""
    inputs = [(1, (1, 0, 0, 0))]
    batch = [read_proof("small_proof_composition.bin"), read_proof("small_proof.bin"), read_proof("small_proof_dup_query.bin")] * 3
    acc, reason = rsv.verify_batch(batch, fixture_cfg("small_proof.bin"), inputs)
    oacc, oreason = ob.verify_batch(batch, fixture_cfg("small_proof.bin"), inputs)
    assert acc.tolist() == oacc.tolist() == [0, 1, 0] * 3
    assert reason.tolist() == oreason.tolist() == [4, 0, 5] * 3


def test_big_shape_reject_fixtures_behind_the_proof_of_work(rsv, knobs):
    """RSV_R_COMPOSITION on a 2^16 / 2^15 proof and RSV_R_DUP_QUERY on the 80-query shape (re-ground fixtures), in
    one mixed batch with genuine proofs around them, under both OODS kernels and both plan kernels."""
    names = ["recursive_proof_16_15_composition.bin", "recursive_proof_16_15.bin", "level1-5_dup_query.bin", "level1-5.bin",
             "recursive_proof_16_15_composition.bin", "level13-1.bin"]
    batch = [read_proof(x) for x in names]
    cfgs = [fixture_cfg(x) for x in names]
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert oacc.tolist() == [0, 1, 0, 1, 0, 1] and oreason.tolist() == [4, 0, 5, 0, 4, 0]
    for oods in ("row", "lane"):
        for plan in ("serial", "parallel"):
            knobs.set("oods_form", oods)
            knobs.set("plan_form", plan)
            acc, reason = rsv.verify_batch(batch, cfgs)
            assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist(), (oods, plan)


@pytest.mark.parametrize("mode", ["row", "lane"])
def test_oods_kernels_row_and_lane(rsv, manifest, knobs, mode):
    """Both OODS kernels (one proof per 16-lane DPP row / one proof per lane): the probe on random samples and the
    whole pipeline on fixtures, wrong public inputs (logup) and the composition reject fixtures."""
    knobs.set("oods_form", mode)
    rng = np.random.default_rng(55)
    n = 200
    sm = rng.integers(0, P, (n, 142, 4), dtype=np.uint32)
    pr = rng.integers(0, P, (n, 26), dtype=np.uint32)
    pr[:, 0], pr[:, 1] = rng.integers(1, 27, n), rng.integers(1, 27, n)
    pr[:4, 0], pr[:4, 1] = [1, 28, 1, 28], [1, 1, 28, 28]
    assert np.array_equal(rsv.oods_eval(sm, pr), ob.oods_eval(sm, pr))
    std = [e for e in manifest if len(e["inputs"]) == 3 and e["expect"] == "ok"]
    batch = [read_proof(e["file"]) for e in std] + [read_proof("recursive_proof_16_15_composition.bin")]
    cfgs = [fixture_cfg(e["file"]) for e in std] + [fixture_cfg("recursive_proof_16_15.bin")]
    acc, reason = rsv.verify_batch(batch, cfgs)
    assert acc.tolist() == [1] * len(std) + [0] and reason.tolist() == [0] * len(std) + [4]
    # 19 public inputs (more than one per lane of a row), all but the genuine three cancelling in pairs is not possible:
    # any extra input breaks the logup sum
    wrong = list(rsv.STANDARD_INPUTS) + [(k, (k, 1, 0, 0)) for k in range(4, 20)]
    acc, reason = rsv.verify_batch(batch[:3], cfgs[:3], wrong)
    oacc, oreason = ob.verify_batch(batch[:3], cfgs[:3], wrong)
    assert acc.tolist() == oacc.tolist() == [0, 0, 0] and reason.tolist() == oreason.tolist() == [3, 3, 3]
    small = [read_proof("small_proof.bin"), read_proof("small_proof_composition.bin")]
    acc, reason = rsv.verify_batch(small, fixture_cfg("small_proof.bin"), [(1, (1, 0, 0, 0))])
    assert acc.tolist() == [1, 0] and reason.tolist() == [0, 4]


def _debug_lib():
    import ctypes
    import torch  # noqa: F401  (one HIP runtime per process: torch's)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recursive-stwo_amd", "csrc", "librsv_hip_count.so")
    if not os.path.exists(path):
        pytest.skip("diagnostic build missing (make -C recursive-stwo_amd/csrc count)")
    lib = ctypes.CDLL(path)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    lib.rsv_debug_plan.restype = ctypes.c_int
    lib.rsv_debug_plan.argtypes = [ctypes.c_uint32] * 5 + [u32p, ctypes.c_int] + [u32p] * 7
    return lib, u32p


def _run_plan(lib, u32p, raw, nq, M, A, B, serial):
    n = raw.shape[0]
    G = max(nq, 4)
    outs = [np.zeros((n, nq), np.uint32), np.zeros((n, nq), np.uint32), np.zeros((n, (M + 1) * G), np.uint32),
            np.zeros((n, 2 * G), np.uint32), np.zeros((n, 32), np.uint32), np.zeros((n, 32), np.uint32), np.zeros((n, 8), np.uint32)]
    rc = lib.rsv_debug_plan(n, nq, M, A, B, raw.ctypes.data_as(u32p), 1 if serial else 0, *[o.ctypes.data_as(u32p) for o in outs])
    assert rc == 0, rc
    return outs


def test_plan_kernels_agree_on_synthetic_query_sets():
    """k_plan_par (bitmask popcounts, one lane per query) against k_plan (serial walk) on shapes no fixture has:
    n_queries 1..128, M up to 30, random / clustered / consecutive positions.  Every table must be identical; with
    duplicate positions (rejected anyway) the sorted list and the DUP flag must agree and every index stay in range."""
    lib, u32p = _debug_lib()
    rng = np.random.default_rng(77)
    shapes = [(1, 5, 4, 3), (4, 6, 5, 5), (5, 9, 8, 7), (8, 26, 25, 24), (10, 24, 23, 23), (11, 30, 29, 28), (16, 22, 21, 20),
              (27, 23, 22, 21), (33, 12, 11, 10), (64, 20, 19, 19), (65, 18, 17, 16), (80, 24, 23, 22), (100, 16, 15, 14),
              (127, 30, 28, 29), (128, 9, 8, 7), (128, 30, 29, 28)]
    for nq, M, A, B in shapes:
        n = 48
        raw = rng.integers(0, 1 << 32, (n, nq), dtype=np.uint64).astype(np.uint32)
        raw[8:16] &= np.uint32((1 << max(M - 3, 1)) - 1) | np.uint32(0xFFFFFFFF << M & 0xFFFFFFFF)   # clustered in the low part
        raw[16:24] = (rng.integers(0, 1 << M, (8, 1), dtype=np.uint64) + np.arange(nq)[None, :]).astype(np.uint32)  # consecutive
        raw[24:32] = (raw[24:32] >> np.uint32(32 - min(M, 8))) << np.uint32(max(M - 8, 0))    # few distinct high prefixes
        raw[32:40, nq // 2:] = raw[32:40, :nq - nq // 2]                                        # duplicates
        ser = _run_plan(lib, u32p, raw, nq, M, A, B, True)
        par = _run_plan(lib, u32p, raw, nq, M, A, B, False)
        qs = ser[0]
        dup = np.array([len(set(r.tolist())) < nq for r in (raw & np.uint32((1 << M) - 1))])
        assert np.array_equal(ser[0], par[0]), (nq, M)                       # sorted positions
        assert np.array_equal(ser[6][:, 0] & 32, par[6][:, 0] & 32)          # R_DUP_QUERY bit
        assert np.array_equal((ser[6][:, 0] & 32) != 0, dup)
        ok = ~dup
        for k, name in enumerate(["q", "qperm", "ent", "fl", "lvl", "wf", "misc"]):
            assert np.array_equal(ser[k][ok], par[k][ok]), (name, nq, M, A, B)
        # in range even for the rejected ones
        G = max(nq, 4)
        ent = par[2].reshape(n, M + 1, G)[:, 1:, :nq]
        sib = (ent >> 16) & 0xFF
        assert ((sib == 0xFF) | (sib < nq)).all() and ((ent & 0xFF) < nq).all()


# ---------------------------------------------------------------- rows a10 / a11 / a12 on their own
def test_oods_eval_probe_matches_oracle(rsv, manifest):
    """rsv_oods_eval (the device function k_oods is made of) against the oracle: every accepting fixture (accumulator
    == expected, bit-identical to the oracle's pair), perturbed samples / challenges (both sides unequal, same values),
    and 300 random sample sets with random log sizes."""
    from tests.test_oracle import oods_params_of
    sm, pr = [], []
    for e in manifest:
        if e["expect"] != "ok":
            continue
        proof = read_proof(e["file"])
        s0, p0 = ob.sampled_values(proof), oods_params_of(proof)
        sm.append(s0); pr.append(p0)
        for k in (0, 57, 110, 133):
            bad = s0.copy(); bad[k, 2] = (int(bad[k, 2]) + 5) % P
            sm.append(bad); pr.append(p0)
        for k in (10, 14, 18, 22):
            badp = p0.copy(); badp[k] = (int(badp[k]) + 1) % P
            sm.append(s0); pr.append(badp)
    rng = np.random.default_rng(31)
    for _ in range(300):
        sm.append(rng.integers(0, P, (142, 4), dtype=np.uint32))
        p = rng.integers(0, P, 26, dtype=np.uint32)
        p[0], p[1] = rng.integers(1, 24), rng.integers(1, 24)
        pr.append(p)
    sm, pr = np.stack(sm), np.stack(pr)
    got, want = rsv.oods_eval(sm, pr), ob.oods_eval(sm, pr)
    assert np.array_equal(got, want)
    n_fix = sum(1 for e in manifest if e["expect"] == "ok")
    for i in range(n_fix):
        base = 9 * i
        assert got[base, :4].tolist() == got[base, 4:].tolist()
        assert all(got[base + k, :4].tolist() != got[base + k, 4:].tolist() for k in range(1, 9))
    with pytest.raises(rsv.RsvError):
        rsv.oods_eval(np.full((1, 142, 4), P, np.uint32), pr[:1])


@pytest.mark.parametrize("name", ["small_proof.bin", "recursive_proof_16_15.bin", "level7-1.bin", "level2-1.bin", "level1-5.bin",
                                  "level10-1.bin", "level13-1.bin"])
@pytest.mark.parametrize("trees", ["paced", "row16"])
def test_query_values_match_oracle(rsv, manifest, knobs, name, trees):
    """Rows a11 / a12 by VALUE (not only by verdict): the DEEP-quotient answers at every column log size, their
    circle-to-line folds, the value entering every inner FRI layer, the value entering the last-layer check and the
    last-layer evaluation, per query, from the verifying pass == the oracle's.  trees: with the row form of the tree kernels
    (what a single proof takes by itself) k_query runs on virtual lanes too — 16 threads per query that split its sums."""
    import torch
    knobs.set("tree_pace", trees)
    knobs.set("query_form", "lane" if trees == "paced" else "row")
    entry = next(e for e in manifest if e["file"] == name)
    proof = read_proof(name)
    lay = ob.proof_layout(proof)
    nq, n_inner = entry["n_queries"], lay["n_inner"]
    M = max(entry["log_size_plonk"] + 1, entry["log_size_poseidon"] + 2) + entry["log_blowup_factor"]
    inputs = entry_inputs(entry)
    want = ob.query_dump(proof, inputs)
    n = 3
    blob, offsets = rsv.pack([proof] * n)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_qv = torch.zeros((n, nq, 4 * (8 + n_inner)), dtype=torch.int32, device=dev)
    ctx = rsv.Context(0)
    # both instantiations of k_query: the chain layout launches the latency form (four columns' loads in flight), the
    # two-stream layout the one that runs beside the trace trees (two)
    for layout in ("on", "off"):
        ctx.set_option("critical_chain", layout)
        d_qv.fill_(-1)
        ctx.verify_hints(d_blob, d_off, n, d_acc, None, cfg=fixture_cfg(name), inputs=inputs, shape=(nq, M, n_inner), d_query_values=d_qv)
        ctx.synchronize()
        assert d_acc.cpu().numpy().tolist() == [1] * n
        got = d_qv.cpu().numpy().view(np.uint32)
        for k in range(n):
            assert np.array_equal(got[k], want), (layout, k)
        ni = n_inner
        assert np.array_equal(got[0][:, 24 + 4 * ni:28 + 4 * ni], got[0][:, 28 + 4 * ni:32 + 4 * ni])  # accept side of RSV_R_FRI_LAST
    ctx.close()


def test_last_layer_check_probe(rsv):
    """RSV_R_FRI_LAST both ways through the device function k_query raises it from: a mutated proof never reaches the
    last-layer comparison (its polynomial is hashed before the proof of work), so the comparison is probed directly —
    equal values pass, any single changed word of the folded value or of a coefficient fails."""
    rng = np.random.default_rng(41)
    for log_n in (0, 2, 7, 8):
        coeffs = rng.integers(0, P, (1 << log_n, 4), dtype=np.uint32)
        xs = rng.integers(0, P, 500, dtype=np.uint32)
        good = ob.line_eval(coeffs, xs)
        assert rsv.last_layer_check(coeffs, xs, good).tolist() == [1] * 500
        bad = good.copy()
        bad[np.arange(500), rng.integers(0, 4, 500)] ^= np.uint32(1)
        bad %= np.uint32(P)
        differs = (bad != good).any(axis=1)
        assert rsv.last_layer_check(coeffs, xs, bad).tolist() == (~differs).astype(np.uint8).tolist() and differs.sum() > 480
        c2 = coeffs.copy(); c2[-1, 3] = (int(c2[-1, 3]) + 1) % P
        assert rsv.last_layer_check(c2, xs, good).sum() == 0  # the top coefficient weighs on every point (weights are products of x's: nonzero w.h.p.)
    with pytest.raises(ValueError):
        rsv.line_eval(np.zeros((3, 4), np.uint32), [1])  # not a power of two


def test_many_distinct_query_counts_in_one_batch(rsv):
    """As many n_queries buckets as one call can carry (16 configurations = RSV_MAX_CFGS = MAX_FUSED since round 3: every
    per-query stage is ONE launch over all of them; until round 2 a launch held 8): the header word n_queries of genuine
    proofs is rewritten to twelve other values and each proof is verified under the matching configuration (16 configurations
    = RSV_MAX_CFGS in one call, 16 distinct n_queries), so every one parses, passes the
    proof of work (n_queries is not part of the transcript before it) and fails in the Merkle stages with its own
    lane geometry; genuine proofs of four shapes sit in between.  Verdict and reason == the oracle's, and a second
    pass through the same context (workspaces already grown) gives the same."""
    import torch
    base = {n: read_proof(n) for n in ("recursive_proof_16_15.bin", "level13-1.bin", "level2-1.bin", "level1-5.bin")}
    batch, cfgs = [], []
    for k in (1, 2, 3, 5, 7, 9, 12, 13, 20, 33, 64, 128):
        name = ("recursive_proof_16_15.bin", "level13-1.bin")[k % 2]
        c0 = fixture_cfg(name)
        b = bytearray(base[name])
        b[4 * 13:4 * 13 + 4] = int(k).to_bytes(4, "little")
        batch.append(bytes(b))
        cfgs.append(type(c0)(c0.pow_bits, c0.log_blowup_factor, c0.log_last_layer_degree_bound, k))
        g = list(base)[len(batch) % 4]
        batch.append(base[g]); cfgs.append(fixture_cfg(g))
    batch = batch * 3
    cfgs = cfgs * 3
    oacc, oreason = ob.verify_batch(batch, cfgs)
    acc, reason = rsv.verify_batch(batch, cfgs)
    diff = np.nonzero((acc != oacc) | (reason != oreason))[0]
    assert diff.size == 0, [(int(i), int(cfgs[i].n_queries), int(reason[i]), int(oreason[i])) for i in diff[:10]]
    assert int(acc.sum()) == 3 * 12  # exactly the genuine proofs
    blob, offsets = rsv.pack(batch)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    n = len(batch)
    ctx = rsv.Context(0)
    pc = ctx.prepare_cfg(cfgs, n)
    for _ in range(2):
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=pc)
        ctx.synchronize()
        assert d_acc.cpu().numpy().tolist() == oacc.tolist() and d_reason.cpu().numpy().tolist() == oreason.tolist()
    ctx.close()


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("top", ["on", "off"])
def test_tree_workgroup_orders_match_oracle(rsv, knobs, order, top):
    """The two workgroup orders of the lane-form tree kernels (RSV_OPT_TREE_ORDER: tree by tree — what production runs — or the
    trees of a workgroup of proofs side by side on one XCD: less HBM traffic, more time), forced on a mixed batch of every chain shape
    (several buckets in one launch, grid x padded to a multiple of 8, a ragged last workgroup) and on one configuration per
    call; tampered copies among them.  Verdicts and reasons == the oracle's."""
    names = ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin", "level7-1.bin",
             "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin", "level13-1.bin"]
    knobs.set("tree_pace", "paced")
    knobs.set("tree_order", order)
    knobs.set("cap_top", top)
    batch, cfgs = [], []
    for k in range(13 * 9 + 4):
        nm = names[k % 13]
        pr = read_proof(nm)
        batch.append(ob.tamper(pr, 2 * k + 1) if k % 5 == 3 else pr)
        cfgs.append(fixture_cfg(nm))
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist() and 60 < int(acc.sum()) < 110
    for nm, n in (("recursive_proof_16_15.bin", 150), ("level1-5.bin", 10), ("level12-1.bin", 33)):
        pr = read_proof(nm)
        one = [ob.tamper(pr, 7 * k) if k % 4 == 1 else pr for k in range(n)]
        a1, r1 = rsv.verify_batch(one, fixture_cfg(nm))
        o1, q1 = ob.verify_batch(one, fixture_cfg(nm))
        assert a1.tolist() == o1.tolist() and r1.tolist() == q1.tolist(), nm


@pytest.mark.parametrize("mid", ["auto", "on"])
def test_cap_kernels_with_every_cap_level(rsv, knobs, mid):
    """The cap kernels (k_cap_mid: a lane per subtree of 4 or 8 nodes; k_cap_top) at every cap level they can meet — 3 (query
    counts 8 .. 15), 4 (16 .. 31), 5 (32 .. 63: a level no fixture has) and 6 (64 .. 128) — and at the workgroup shapes those
    counts give (two 128-query proofs per workgroup, four of 63, 25 of 10): the header word n_queries of genuine proofs is
    rewritten and each proof verified under the matching configuration, so it parses, passes the proof of work and fails in
    the Merkle stages with its own geometry; genuine proofs of four shapes in between.  Verdict and reason == the oracle's,
    with the hand-over forced for every bucket and chosen by fill."""
    knobs.set("tree_pace", "paced")
    knobs.set("cap_top", "on")
    knobs.set("cap_mid", mid)
    base = {n: read_proof(n) for n in ("recursive_proof_16_15.bin", "level13-1.bin", "level2-1.bin", "level1-5.bin")}
    batch, cfgs = [], []
    for k in (4, 6, 9, 15, 20, 31, 32, 33, 40, 63, 64, 100, 128):
        name = ("recursive_proof_16_15.bin", "level13-1.bin", "level1-5.bin")[k % 3]
        c0 = fixture_cfg(name)
        for rep in range(5):  # several proofs per workgroup, tampered ones among them
            b = bytearray(ob.tamper(base[name], 3 * k + rep) if rep == 3 else base[name])
            b[4 * 13:4 * 13 + 4] = int(k).to_bytes(4, "little")
            batch.append(bytes(b))
            cfgs.append(type(c0)(c0.pow_bits, c0.log_blowup_factor, c0.log_last_layer_degree_bound, k))
        g = list(base)[len(batch) % 4]
        batch.append(base[g]); cfgs.append(fixture_cfg(g))
    oacc, oreason = ob.verify_batch(batch, cfgs)
    for lo in range(0, len(batch), 30):  # (16 configurations per call: RSV_MAX_CFGS)
        acc, reason = rsv.verify_batch(batch[lo:lo + 30], cfgs[lo:lo + 30])
        diff = np.nonzero((acc != oacc[lo:lo + 30]) | (reason != oreason[lo:lo + 30]))[0]
        assert diff.size == 0, [(int(i), int(cfgs[lo + i].n_queries), int(reason[i]), int(oreason[lo + i])) for i in diff[:10]]
    assert int(oacc.sum()) == 13  # exactly the genuine proofs
    # one configuration per call (the device-side slot order sizes its launch for the deepest trees the parser admits)
    for k in (33, 100):
        one = [b for b, c in zip(batch, cfgs) if c.n_queries == k]
        a1, r1 = rsv.verify_batch(one, cfgs[[c.n_queries for c in cfgs].index(k)])
        o1, q1 = ob.verify_batch(one, cfgs[[c.n_queries for c in cfgs].index(k)])
        assert a1.tolist() == o1.tolist() and r1.tolist() == q1.tolist(), k


def test_mixed_batch_split_into_several_launch_groups(rsv, monkeypatch):
    """A mixed batch whose per-query workspaces exceed the budget: the launcher cuts it into several groups of
    (bucket, slot range) entries, some buckets split across groups.  Verdicts == those of the unconstrained run ==
    the oracle's on the distinct inputs."""
    import torch
    names = ["level1-5.bin", "recursive_proof_16_15.bin", "level2-1.bin", "level8-1.bin", "level13-1.bin", "level10-1.bin"]
    proofs = {x: read_proof(x) for x in names}
    n = 2400
    batch, cfgs, want_names = [], [], []
    for i in range(n):
        x = names[i % len(names)]
        batch.append(ob.tamper(proofs[x], i) if i % 13 == 4 else proofs[x])
        cfgs.append(fixture_cfg(x))
    tam = [i for i in range(n) if i % 13 == 4]
    oacc, oreason = ob.verify_batch([batch[i] for i in tam], [cfgs[i] for i in tam])
    want_acc, want_reason = np.ones(n, np.uint8), np.zeros(n, np.uint8)
    want_acc[tam], want_reason[tam] = oacc, oreason
    blob, offsets = rsv.pack(batch)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    for budget in (64, 8192):
        ctx = rsv.Context(0)
        ctx.set_option("ws_budget_mb", budget)
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=ctx.prepare_cfg(cfgs, n))
        ctx.synchronize()
        assert np.array_equal(d_acc.cpu().numpy(), want_acc), budget
        assert np.array_equal(d_reason.cpu().numpy(), want_reason), budget
        ctx.close()


@pytest.mark.parametrize("mode", ["row", "lane"])
def test_qconst_kernels_row_and_lane(rsv, manifest, knobs, mode):
    """Both forms of the query-independent quotient constants (one proof per 16-lane row / per lane): every fixture of
    the standard inputs accepts, tampers keep the oracle's reasons, and the per-query quotient / fold values — which
    consume every alpha power and every summed line coefficient — equal the oracle's for three size-group layouts
    (A > B, A == B, A < B)."""
    import torch
    knobs.set("qconst_form", mode)
    std = [e for e in manifest if len(e["inputs"]) == 3 and e["expect"] == "ok"]
    batch = [read_proof(e["file"]) for e in std]
    cfgs = [fixture_cfg(e["file"]) for e in std]
    batch += [ob.tamper(batch[k], 7 + k) for k in range(len(std))]
    cfgs += cfgs
    acc, reason = rsv.verify_batch(batch, cfgs)
    oacc, oreason = ob.verify_batch(batch, cfgs)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    assert acc[:len(std)].tolist() == [1] * len(std)
    dev = torch.device("cuda:0")
    for name in ("recursive_proof_16_15.bin", "level7-1.bin", "small_proof.bin"):
        entry = next(e for e in manifest if e["file"] == name)
        proof = read_proof(name)
        nq, n_inner = entry["n_queries"], ob.proof_layout(proof)["n_inner"]
        M = max(entry["log_size_plonk"] + 1, entry["log_size_poseidon"] + 2) + entry["log_blowup_factor"]
        blob, offsets = rsv.pack([proof])
        d_blob = torch.from_numpy(blob.copy()).to(dev)
        d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
        d_acc = torch.zeros(1, dtype=torch.uint8, device=dev)
        d_qv = torch.zeros((1, nq, 4 * (8 + n_inner)), dtype=torch.int32, device=dev)
        ctx = rsv.Context(0)
        ctx.verify_hints(d_blob, d_off, 1, d_acc, None, cfg=fixture_cfg(name), inputs=entry_inputs(entry), shape=(nq, M, n_inner),
                         d_query_values=d_qv)
        ctx.synchronize()
        assert int(d_acc.item()) == 1
        assert np.array_equal(d_qv.cpu().numpy().view(np.uint32)[0], ob.query_dump(proof, entry_inputs(entry))), name
        ctx.close()


FLOW_SHAPES = ["small_proof.bin", "recursive_proof_16_15.bin", "level7-1.bin", "level2-1.bin", "level1-5.bin", "level9-1.bin", "level13-1.bin"]


@pytest.mark.parametrize("trees", ["paced", "row16"])
@pytest.mark.parametrize("form", ["row", "lane", "lane, every lane walks to the root"])
@pytest.mark.parametrize("name", FLOW_SHAPES)
def test_poseidon_flow_matches_oracle(rsv, manifest, knobs, name, form, trees):
    """SURVEY 8f.1, second half: the PoseidonFlow records the verifying pass writes (rsv_hints_out::d_flow) == the
    oracle's, which runs the reference's per-path verifiers in the circuit's invocation order with a recorder on its
    permutation (oracle/rsv_oracle.c: rsvo_poseidon_flow).  Seven shapes: n_queries 8 / 11 / 16 / 27 / 80, equal and
    unequal column log sizes (level7-1: lp = lq, two column levels instead of three).  Every record is an invocation
    (perm(inputs) = outputs, checked on the GPU's own permutation entry point) and the count is the shape's.  Both forms of
    the transcript kernel write the channel's records: one proof per 16-lane row (small batches) and one proof per lane."""
    # default: the top-of-tree cap hashes a node several queries share ONCE and writes every query's record from there;
    # flow_cap = off: every lane hashes (and records) its whole path itself
    # trees: the flow-writing tree kernels in the lane form and in the row form on virtual lanes (16 threads write each
    # record; what a three-proof call takes by itself; the 80-query shape falls back to the lane form)
    knobs.set("transcript_form", form.split(",")[0])
    knobs.set("flow_cap", "off" if "," in form else "auto")
    knobs.set("tree_pace", trees)
    entry = next(e for e in manifest if e["file"] == name)
    proof, inputs, cfg = read_proof(name), entry_inputs(entry), fixture_cfg(name)
    want = ob.poseidon_flow(proof, inputs)
    cnt = rsv.poseidon_flow_count(entry["log_size_plonk"], entry["log_size_poseidon"], cfg)
    assert cnt == len(want)
    flow, swap, count, acc, reason = rsv.poseidon_flow([proof, ob.tamper(proof, 3), proof], cfg, cnt + 5, inputs)
    assert acc.tolist() == [1, 0, 1] and count.tolist() == [cnt, cnt, cnt]
    for k in (0, 2):
        assert np.array_equal(flow[k, :cnt], want[:, :32]), int(np.nonzero((flow[k, :cnt] != want[:, :32]).any(axis=1))[0][0])
        assert np.array_equal(swap[k, :cnt], want[:, 32].astype(np.uint8))
        assert not flow[k, cnt:].any() and not swap[k, cnt:].any()
    left, right, out = flow[0, :cnt, 0:8], flow[0, :cnt, 8:16], flow[0, :cnt, 16:32]
    state = np.where(swap[0, :cnt, None] == 1, np.concatenate([right, left], 1), np.concatenate([left, right], 1))
    assert np.array_equal(rsv.poseidon2_permute(state), out)
    # a stride one record too small: the proof is still verified, nothing is written, its count says 0
    flow1, swap1, count1, acc1, _ = rsv.poseidon_flow([proof], cfg, cnt - 1, inputs)
    assert acc1.tolist() == [1] and count1.tolist() == [0] and not flow1.any() and not swap1.any()


def test_poseidon_flow_mixed_batch_and_count_pin(rsv, manifest):
    """One mixed batch (five shapes, a tampered and a garbage proof between them): per-proof counts and records, and the
    count pin of tests/test_oracle.py::test_poseidon_flow_count_predicts_next_level from the GPU's side: the flow of
    level K's verification, padded, is the Poseidon trace whose log size level K+1's header carries."""
    names = ["recursive_proof_16_15.bin", "level8-1.bin", "level12-1.bin", "level3-1.bin", "level10-1.bin"]
    proofs = [read_proof(x) for x in names]
    batch = proofs[:2] + [ob.tamper(proofs[0], 9), b"\x00" * 4000] + proofs[2:]
    cfgs = [fixture_cfg(x) for x in names[:2]] + [fixture_cfg(names[0])] * 2 + [fixture_cfg(x) for x in names[2:]]
    stride = 6000
    flow, swap, count, acc, reason = rsv.poseidon_flow(batch, cfgs, stride)
    assert acc.tolist() == [1, 1, 0, 0, 1, 1, 1] and count[3] == 0 and reason[3] == 1
    for k, x in zip([0, 1, 4, 5, 6], names):
        want = ob.poseidon_flow(read_proof(x))
        assert count[k] == len(want) and np.array_equal(flow[k, :len(want)], want[:, :32]) and np.array_equal(swap[k, :len(want)], want[:, 32])
    nxt = {"recursive_proof_16_15.bin": (5, "level1-5.bin"), "level8-1.bin": (1, "level9-1.bin"), "level12-1.bin": (1, "level13-1.bin"),
           "level3-1.bin": (5, "level4-5.bin"), "level10-1.bin": (1, "level11-1.bin")}
    for k, x in zip([0, 1, 4, 5, 6], names):
        mult, dst = nxt[x]
        assert ob.flow_log_size(mult * int(count[k])) == next(e for e in manifest if e["file"] == dst)["log_size_poseidon"]


def _reshaped(proof: bytes, lp: int, lq: int) -> bytes:
    """The fixture re-serialized under another pair of component log sizes: header words changed, the list of inner FRI
    layers cut or padded to the length the parser derives from them.  Parses; cannot verify."""
    d = ob.split_variable_part(proof)
    head = d["head"].copy()
    head[0], head[1] = lp, lq
    d["head"] = head
    b, last = int(head[11]), int(head[12])
    n_inner = max(lp + 1, lq + 2) + b - 1 - (last + b)
    layers = d["layers"]
    d["layers"] = [layers[0]] + [layers[1 + (i % (len(layers) - 1))] for i in range(n_inner)]
    return ob.join_variable_part(d)


@pytest.mark.parametrize("order", ["device", "host"])
def test_slot_order_on_the_device_and_on_the_host(rsv, knobs, order):
    """A batch under ONE configuration takes no host round trip: the slot order by shape class is made by three small
    kernels (csrc/k_parse.hpp: k_classify / k_class_offsets / k_scatter_ids) and the tables are sized for the deepest
    trees the parser admits.  The same batch through the host-side bucketing of rounds 1-2 (device_order = host) and
    through the device path must give the oracle's verdicts: the four standard-configuration fixtures interleaved
    (three geometries), tampered copies, garbage and truncated buffers (the unparsed class), and 150 parseable proofs of
    ~90 DISTINCT shapes (more than the 63 entries of the class table: the overflow class)."""
    knobs.set("device_order", order)
    names = ["recursive_proof_16_15.bin", "level3-1.bin", "level6-1.bin", "level7-1.bin"]
    proofs = [read_proof(x) for x in names]
    cfg = fixture_cfg(names[0])
    batch = []
    for i in range(600):
        pr = proofs[i % 4]
        batch.append(ob.tamper(pr, i) if i % 7 == 3 else (pr[: 4000 + 4 * i] if i % 53 == 9 else pr))
    shapes = [(lp, lq) for lp in range(9, 24) for lq in range(9, 22) if 14 <= max(lp + 1, lq + 2) + 5 <= 30][:90]
    assert len(set(shapes)) > 63
    for k in range(150):
        lp, lq = shapes[k % len(shapes)]
        batch.append(_reshaped(proofs[k % 4], lp, lq))
    batch += [b"", b"\x00" * 8000, proofs[0]]
    acc, reason = rsv.verify_batch(batch, cfg)
    oacc, oreason = ob.verify_batch(batch, cfg)
    assert acc.tolist() == oacc.tolist() and reason.tolist() == oreason.tolist()
    assert int(acc.sum()) > 400 and acc[-1] == 1 and 1 in reason.tolist() and 2 in reason.tolist()


def test_graph_replay_of_repeated_calls(rsv):
    """RSV_OPT_GRAPH (experiment): a call repeated with identical arguments is captured into a HIP graph on its second
    sighting and replayed afterwards.  The replay reads the buffers as they are THEN: the blob is tampered in place
    between calls and the verdicts must follow; a call with other arguments drops the graph."""
    import torch
    dev = torch.device("cuda:0")
    proof = read_proof("recursive_proof_16_15.bin")
    n = 1500
    blob, offsets = rsv.pack([proof] * n)
    d_blob = torch.from_numpy(blob.copy()).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx = rsv.Context(0)
    ctx.set_option("graph", "on")
    cfg = rsv.PreparedCfg([fixture_cfg("recursive_proof_16_15.bin")])
    want = np.ones(n, np.uint8)
    for rep in range(6):
        if rep >= 2:  # tamper one more proof in place: plain call, capture, replays all see the bytes of the moment
            k = 100 * rep + 7
            d_blob[int(offsets[k]) + 5000] ^= 1
            want[k] = 0
        d_acc.fill_(9)
        ctx.verify_batch(d_blob, d_off, n, d_acc, d_reason, cfg=cfg)
        ctx.synchronize()
        assert np.array_equal(d_acc.cpu().numpy(), want), rep
    d_acc2 = torch.zeros(n - 1, dtype=torch.uint8, device=dev)  # other arguments: the cached graph is dropped
    ctx.verify_batch(d_blob, d_off, n - 1, d_acc2, None, cfg=cfg)
    ctx.synchronize()
    assert np.array_equal(d_acc2.cpu().numpy(), want[:n - 1])
    ctx.close()
