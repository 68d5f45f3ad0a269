"""rsv_witness_program_build + rsv_witness_eval (`-m gpu`): the library's own mirror of the circuit's gadgets (C++, fed with
the GPU's hints of a template proof) must write the program the CPU oracle's restatement of the circuit derives, byte for
byte; and the recursion circuit's `variables` vector, evaluated on the GPU for a batch, must equal the values the oracle's
gadgets compute when they are run — proof by proof, Python integers, CPU oracle's hints — on the same proofs
(oracle/recursion_circuit, itself pinned to the reference's fixtures by tests/test_recursion_circuit.py)."""
import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.conftest import fixture_cfg, load_manifest, read_proof

pytestmark = pytest.mark.gpu
MAN = {e["file"]: e for e in load_manifest()}


def _inputs(name):
    return [(i, tuple(v)) for i, v in MAN[name]["inputs"]]


def _oracle_variables(name, copies=1):
    from oracle import recursion_circuit as rc
    c, d, _ = rc.build_circuit(read_proof(name), ob, _inputs(name), copies)
    c.check_arithmetics()
    return np.array(c.variables, dtype=np.uint32), c, d


@pytest.mark.parametrize("template,other", [("level10-1.bin", "level11-1.bin"), ("level2-1.bin", "level5-1.bin"),
                                            ("level1-5.bin", "level4-5.bin"),  # 80 queries, 300 000 variables
                                            ("level7-1.bin", "level7-1.bin"),  # both components of one log size
                                            ("small_proof.bin", "small_proof.bin")])  # one public input, last layer of 4
def test_witness_of_a_batch_matches_the_gadgets(rsv, template, other):
    """Program from `template` (GPU hints), batch = [template, other, a tampered copy, a proof of another shape, other]:
    the rows of the two valid proofs of the shape equal the gadgets' `variables`, the tampered and the foreign proof are
    flagged; the program equals the one extracted from the oracle's run (same instructions, same levels)."""
    cfg = fixture_cfg(template)
    wp = rsv.WitnessProgram.build(read_proof(template), cfg, _inputs(template))
    prog = wp.export()
    want_t, c, d = _oracle_variables(template)
    want_o, c_other, _ = _oracle_variables(other)
    from oracle import recursion_circuit as rc
    ref = rc.program.extract(c, d)
    assert prog.n_vars == ref.n_vars and prog.shape == ref.shape
    assert np.array_equal(prog.instr, ref.instr) and np.array_equal(prog.level_offsets, ref.level_offsets) and np.array_equal(prog.flow_wires, ref.flow_wires)
    foreign = "level12-1.bin" if template != "level12-1.bin" else "level9-1.bin"
    batch = [read_proof(template), read_proof(other), ob.tamper(read_proof(template), 7), read_proof(foreign), read_proof(other)]
    variables, accept, reason, flow, swap = rsv.witness(batch, wp, _inputs(template), with_flow=True)
    assert accept.tolist() == [1, 1, 0, 0, 1] and reason[2] != 0
    # the PoseidonFlow that comes with it: the hashes == the CPU checker's record for record; its wire indices (constants
    # of the shape, Program.flow_wires) point at Poseidon gates whose two operands ARE the entry's hash in this proof's
    # variables (check_poseidon_invocations, plonk_with_poseidon.rs:468-490), and the swap address holds the swap bit
    want_flow = ob.poseidon_flow(read_proof(other), _inputs(other))
    assert np.array_equal(flow[1], want_flow[:, :32]) and np.array_equal(swap[1], want_flow[:, 32].astype(np.uint8))
    at = np.empty(prog.n_vars, np.int64)
    at[prog.instr[:, 1]] = np.arange(prog.n_vars)
    P_ = rsv.witness_program
    checked = 0
    for k in range(0, prog.shape["flow_count"], 7):
        for j in range(4):
            w = int(prog.flow_wires[k, j])
            if w:
                op, _, a, b = (int(x) for x in prog.instr[at[w], :4])
                assert op == P_.MUL
                assert np.array_equal(np.concatenate([variables[1][a], variables[1][b]]), flow[1][k, 8 * j:8 * j + 8]), (k, j)
                checked += 1
        addr = int(prog.flow_wires[k, 4])
        assert variables[1][addr].tolist() == [int(swap[1][k]), 0, 0, 0] if addr else swap[1][k] == 0
    assert checked > prog.shape["flow_count"] // 7
    assert np.array_equal(variables[0], want_t)
    assert np.array_equal(variables[1], want_o) and np.array_equal(variables[4], want_o)
    # and the vector is a witness: every gate of ITS circuit holds on it (check_arithmetics, plonk_with_poseidon.rs:302-343;
    # the gate list of `other`, not the template's: a few gate constants of the reference follow the witness, gadgets.pm_select)
    c_other.variables = [tuple(int(x) for x in v) for v in variables[1]]
    c_other.check_arithmetics()
    wp.close()


def test_builder_writes_the_oracles_program_for_every_fixture_shape(rsv, manifest):
    """rsv_witness_program_build (the library's C++ mirror of the gadgets, GPU hints) against the oracle's Python restatement
    (CPU oracle's hints) on every Poseidon-channel fixture of the reference — 15 proofs, 11 distinct shapes, n_queries 8 … 80,
    last layers of 4 … 256 coefficients, one and three public inputs: the same instructions in the same levels and the same
    flow wires, byte for byte; and each program evaluates its own template to the oracle's `variables`."""
    from oracle import recursion_circuit as rc
    seen = set()
    for e in manifest:
        if e.get("expect") != "ok" or "struct" in e:
            continue
        name = e["file"]
        proof, inputs = read_proof(name), _inputs(name)
        wp = rsv.WitnessProgram.build(proof, fixture_cfg(name), inputs)
        prog = wp.export()
        c, d, _ = rc.build_circuit(proof, ob, inputs)
        ref = rc.program.extract(c, d)
        assert prog.shape == ref.shape and prog.n_vars == ref.n_vars, name
        assert np.array_equal(prog.instr, ref.instr) and np.array_equal(prog.level_offsets, ref.level_offsets), name
        assert np.array_equal(prog.flow_wires, ref.flow_wires), name
        key = tuple(sorted(prog.shape.items()))
        if key not in seen:  # one evaluation per shape
            seen.add(key)
            variables, accept, _ = rsv.witness([proof], wp, inputs)
            assert accept[0] == 1 and np.array_equal(variables[0], np.array(c.variables, dtype=np.uint32)), name
        wp.close()
    assert len(seen) >= 10


def _oracle_rows(c):
    return np.stack([np.array(x, dtype=np.int64) % 0x7FFFFFFF for x in (c.a_wire, c.b_wire, c.c_wire, c.op, c.poseidon_wire, c.enforce_c_m31)], axis=1)


def test_gate_list_is_the_oracles(rsv, manifest):
    """rsv_witness_program_gates: the Plonk rows the library's gadgets left (wires, op, Poseidon wire, enforce_c_m31) equal
    the oracle's constraint system row for row on every fixture; and for ANOTHER proof of a shape the rows whose `op`
    follows the witness (CirclePointM31Var::select), set from that proof's GPU variables, give that proof's own gate list."""
    from oracle import recursion_circuit as rc
    for e in manifest:
        if e.get("expect") != "ok" or "struct" in e:
            continue
        name = e["file"]
        wp = rsv.WitnessProgram.build(read_proof(name), fixture_cfg(name), _inputs(name))
        rows, ops = wp.gates()
        c, _, _ = rc.build_circuit(read_proof(name), ob, _inputs(name))
        assert np.array_equal(rows, _oracle_rows(c)), name
        assert len(ops) > 0 and (rows[ops[:, 0], 1] == 0).all()
        wp.close()
    wp = rsv.WitnessProgram.build(read_proof("level10-1.bin"), fixture_cfg("level10-1.bin"))
    variables, accept, _ = rsv.witness([read_proof("level11-1.bin")], wp)
    template_rows, _ = wp.gates()
    rows, _ = wp.gates(variables[0])
    c, _, _ = rc.build_circuit(read_proof("level11-1.bin"), ob)
    assert accept[0] == 1 and np.array_equal(rows, _oracle_rows(c)) and not np.array_equal(rows, template_rows)
    wp.close()


def test_everything_the_next_prover_reads_comes_out_of_the_library(rsv):
    """generate_plonk_with_poseidon_circuit + populate_logup_arguments (plonk_with_poseidon.rs:345-629) fed with the library's
    outputs ONLY — variables and flow from the GPU, gate list and flow wires from the built program — give the 22 Plonk
    columns the next fixture proves (the oracle package is used for the padding / multiplicity / interpolation arithmetic,
    not for a single value of the circuit)."""
    from oracle import recursion_circuit as rc
    from oracle.recursion_circuit import trace as T
    pin = next(p for p in _pins() if p["src"] == "level11-1.bin")
    orders = [tuple(tuple(x) for x in o) for o in pin["shift_orders"]]
    walks = [(1 if o[0] == (-1, 0) else 0) | (2 if o[1] == (-1, 0) else 0) for o in orders]
    src = pin["src"]
    wp = rsv.WitnessProgram.build(read_proof("level10-1.bin"), fixture_cfg(src), set_walks=walks)  # another proof as the template
    variables, accept, _, flow, swap = rsv.witness([read_proof(src)], wp, with_flow=True)
    rows, _ = wp.gates(variables[0])
    wires = wp.export().flow_wires
    c = rc.cs.ConstraintSystem()
    c.variables = [tuple(int(x) for x in v) for v in variables[0]]
    c.a_wire, c.b_wire, c.c_wire, c.op, c.poseidon_wire, c.enforce_c_m31 = (rows[:, k].tolist() for k in range(6))
    c.flow = [((int(w[0]), None), (int(w[1]), None), (int(w[2]), None), (int(w[3]), None), int(w[4]), bool(sw)) for w, sw in zip(wires, swap[0])]
    c.check_arithmetics()
    nxt = read_proof(pin["dst"])
    tr = ob.transcript_raw(nxt)
    oods = (tuple(int(x) for x in tr[20:24]), tuple(int(x) for x in tr[24:28]))
    want = rc.parse_proof(nxt).sampled_values
    lp = int(np.frombuffer(nxt[:4], np.uint32)[0])
    assert accept[0] == 1 and T.pad(c) == 1 << lp
    pre, trace = T.plonk_columns(c)
    pe = T.PointEvaluator(lp, oods)
    for k, name in enumerate(T.PREPROCESSED):
        assert pe.eval(pre[name]) == want[0][k][0], name
    for k in range(12):
        assert pe.eval(trace[k]) == want[1][k][0], k
    wp.close()


def test_builder_follows_the_four_hashset_walks(rsv):
    """set_walks: for each of the four orders in which a run of the reference can walk its two HashSets (and a mix of
    them over three copies) the builder writes the program of the oracle's circuit under the same orders; the four programs
    differ from each other, their value multisets do not."""
    from oracle import recursion_circuit as rc
    name = "level12-1.bin"
    proof, inputs, cfg = read_proof(name), _inputs(name), fixture_cfg(name)
    order_of = {0: ((0, -1), (0, -1)), 1: ((-1, 0), (0, -1)), 2: ((0, -1), (-1, 0)), 3: ((-1, 0), (-1, 0))}
    progs = []
    for walks in ([0], [1], [2], [3], [2, 0, 3]):
        wp = rsv.WitnessProgram.build(proof, cfg, inputs, copies=len(walks), set_walks=walks)
        prog = wp.export()
        c, d, _ = rc.build_circuit(proof, ob, inputs, len(walks), [order_of[w] for w in walks])
        ref = rc.program.extract(c, d, len(walks))
        assert np.array_equal(prog.instr, ref.instr) and np.array_equal(prog.level_offsets, ref.level_offsets), walks
        assert np.array_equal(prog.flow_wires, ref.flow_wires), walks
        if len(walks) == 1:
            variables, accept, _ = rsv.witness([proof], wp, inputs)
            assert accept[0] == 1 and np.array_equal(variables[0], np.array(c.variables, dtype=np.uint32))
            progs.append((prog, variables[0]))
        wp.close()
    for k in range(1, 4):
        assert not np.array_equal(progs[0][0].instr, progs[k][0].instr)
        assert np.array_equal(np.sort(progs[0][1].view("u4,u4,u4,u4").ravel()), np.sort(progs[k][1].view("u4,u4,u4,u4").ravel()))
    with pytest.raises(ValueError):
        rsv.WitnessProgram.build(proof, cfg, inputs, copies=2, set_walks=[0])


def test_witness_of_five_copies_in_one_circuit(rsv):
    """examples/multi-proofs verifies recursive_proof_16_15.bin five times in one circuit (main.rs:64, 173-196):
    the program of that circuit (about 330 000 variables; its 291 870 Plonk rows are the level1 fixture's 2^19)."""
    name = "recursive_proof_16_15.bin"
    wp = rsv.WitnessProgram.build(read_proof(name), fixture_cfg(name), _inputs(name), copies=5)
    want, c, d = _oracle_variables(name, 5)
    from oracle import recursion_circuit as rc
    prog, ref = wp.export(), rc.program.extract(c, d, 5)
    assert np.array_equal(prog.instr, ref.instr) and np.array_equal(prog.flow_wires, ref.flow_wires) and prog.shape["copies"] == 5
    variables, accept, _ = rsv.witness([read_proof(name)] * 3, wp, _inputs(name))
    assert accept.tolist() == [1, 1, 1]
    for k in range(3):
        assert np.array_equal(variables[k], want)
    wp.close()


def _pins():
    import json
    import os
    from tests.conftest import GOLDEN
    with open(os.path.join(GOLDEN, "recursion_circuit_pins.json")) as f:
        return json.load(f)["pairs"]


@pytest.mark.parametrize("pin", _pins(), ids=lambda p: f"{p['src']}x{p['multiplier']}")
def test_gpu_witness_is_what_the_next_fixture_proves(rsv, pin):
    """The GPU's vector against the REFERENCE directly, for all 14 consecutive fixture pairs: fixture K+1 is the proof of the
    circuit that verifies fixture K (`multiplier` times), so the 12 trace columns a_val / b_val / c_val of that circuit —
    the GPU's `variables` of K read through the wires — interpolated and evaluated at K+1's OODS point, are the sampled
    values K+1 carries.  The library's builder is told which way the reference's run walked its two HashSets for that
    fixture (set_walks, from the recorded pins); the gate list comes from the oracle's run with the same walks."""
    from oracle import recursion_circuit as rc
    from oracle.recursion_circuit import trace as T
    src, mult = pin["src"], pin["multiplier"]
    orders = [tuple(tuple(x) for x in o) for o in pin["shift_orders"]]
    walks = [(1 if o[0] == (-1, 0) else 0) | (2 if o[1] == (-1, 0) else 0) for o in orders]
    wp = rsv.WitnessProgram.build(read_proof(src), fixture_cfg(src), _inputs(src), copies=mult, set_walks=walks)
    variables, accept, _ = rsv.witness([read_proof(src)], wp, _inputs(src))
    assert accept[0] == 1
    c, _, _ = rc.build_circuit(read_proof(src), ob, _inputs(src), mult, orders)  # for the gate list: the values come from the GPU
    assert len(c.variables) == wp.n_vars
    c.variables = [tuple(int(x) for x in v) for v in variables[0]]
    nxt = read_proof(pin["dst"])
    tr = ob.transcript_raw(nxt)
    oods = (tuple(int(x) for x in tr[20:24]), tuple(int(x) for x in tr[24:28]))
    want = rc.parse_proof(nxt).sampled_values
    lp = int(np.frombuffer(nxt[:4], np.uint32)[0])
    assert T.pad(c) == 1 << lp
    _, trace = T.plonk_columns(c)
    pe = T.PointEvaluator(lp, oods)
    for k in range(12):
        assert pe.eval(trace[k]) == want[1][k][0], k
    wp.close()


@pytest.mark.parametrize("pin", _pins(), ids=lambda p: f"{p['src']}x{p['multiplier']}")
def test_gpu_flow_is_the_next_fixtures_poseidon_component(rsv, pin):
    """The same for the PoseidonFlow the call returns, all 14 pairs: the GPU's records (hashes + swap bits; one copy's,
    repeated per copy) with the program's wire indices, padded and laid out six rows per invocation, ARE the Poseidon
    component of the next fixture — all 40 preprocessed and 48 trace columns evaluate to its sampled values at its OODS point."""
    import ctypes
    from oracle import recursion_circuit as rc
    from oracle.recursion_circuit import trace as T
    src, mult = pin["src"], pin["multiplier"]
    orders = [tuple(tuple(x) for x in o) for o in pin["shift_orders"]]
    walks = [(1 if o[0] == (-1, 0) else 0) | (2 if o[1] == (-1, 0) else 0) for o in orders]
    wp = rsv.WitnessProgram.build(read_proof(src), fixture_cfg(src), _inputs(src), copies=mult, set_walks=walks)
    _, accept, _, flow, swap = rsv.witness([read_proof(src)], wp, _inputs(src), with_flow=True)
    wires = wp.export().flow_wires
    count = flow.shape[1]
    assert accept[0] == 1 and len(wires) == mult * count
    recs = [((int(w[0]), tuple(int(x) for x in f[0:8])), (int(w[1]), tuple(int(x) for x in f[8:16])), (int(w[2]), tuple(int(x) for x in f[16:24])),
             (int(w[3]), tuple(int(x) for x in f[24:32])), int(w[4]), bool(sw))
            for w, f, sw in zip(wires, np.tile(flow[0], (mult, 1)), np.tile(swap[0], mult))]
    for _ in range(len(recs), max(32, -(-len(recs) // 16) * 16)):  # PlonkWithPoseidonConstraintSystem::pad
        recs.append(((0, None), (0, None), (0, None), (0, None), 0, False))
    ob.lib.rsvo_round_constants.restype = ctypes.POINTER(ctypes.c_uint32)
    r = [ob.lib.rsvo_round_constants(k) for k in range(3)]
    rcs = ([[int(r[0][16 * a + i]) for i in range(16)] for a in range(4)], [int(r[1][i]) for i in range(14)],
           [[int(r[2][16 * a + i]) for i in range(16)] for a in range(4)])
    nxt = read_proof(pin["dst"])
    tr = ob.transcript_raw(nxt)
    oods = (tuple(int(x) for x in tr[20:24]), tuple(int(x) for x in tr[24:28]))
    want = rc.parse_proof(nxt).sampled_values
    lq = int(np.frombuffer(nxt[:8], np.uint32)[1])
    qpre, qtr = T.poseidon_columns(recs, rcs, lq, padding_hash=([0] * 8,))
    pe = T.PointEvaluator(lq, oods)
    assert all(pe.eval(qpre[k]) == want[0][10 + k][0] for k in range(40))
    assert all(pe.eval(qtr[k]) == want[1][12 + k][0] for k in range(48))
    wp.close()


def test_one_context_many_shapes_and_batch_sizes(rsv):
    """Soak of the scratch handling: ONE context, programs of five shapes used in turn, batch sizes 1 … 150 in random order,
    a random subset of every batch tampered (and one proof of a foreign shape): accept flags as rsv_verify_batch's, every
    accepted row equal to the oracle's vector for that proof — nothing of an earlier batch or shape leaks into a later one."""
    import torch
    rng = np.random.default_rng(7)
    dev = torch.device("cuda:0")
    ctx = rsv.Context(0)
    pairs = [("level10-1.bin", "level11-1.bin"), ("level12-1.bin", "level12-1.bin"), ("level2-1.bin", "level5-1.bin"), ("small_proof.bin", "small_proof.bin"),
             ("level7-1.bin", "level7-1.bin")]
    progs, wants = {}, {}
    for a, b in pairs:
        progs[a] = rsv.WitnessProgram.build(read_proof(a), fixture_cfg(a), _inputs(a))
        for x in (a, b):
            wants[x] = _oracle_variables(x)[0]
    for it in range(12):
        a, b = pairs[int(rng.integers(len(pairs)))]
        wp, n = progs[a], int(rng.choice([1, 2, 3, 17, 64, 65, 150]))
        names = [a if rng.random() < 0.5 else b for _ in range(n)]
        batch = [read_proof(x) for x in names]
        bad = set(int(i) for i in np.nonzero(rng.random(n) < 0.3)[0])
        for i in bad:
            batch[i] = ob.tamper(batch[i], 1000 * it + i)
        foreign = int(rng.integers(n)) if n > 2 else None
        if foreign is not None:
            batch[foreign] = read_proof("level9-1.bin" if a != "level9-1.bin" else "level8-1.bin")
        blob, offsets = rsv.pack(batch)
        d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
        d_vars = torch.full((n, wp.n_vars, 4), -1, dtype=torch.int32, device=dev)
        d_acc = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=_inputs(a))
        ctx.synchronize()
        acc, got = d_acc.cpu().numpy(), d_vars.cpu().numpy().view(np.uint32)
        for i in range(n):
            good = i not in bad and i != foreign
            assert acc[i] == (1 if good else 0), (it, a, n, i)
            if good:
                assert np.array_equal(got[i], wants[names[i]]), (it, a, n, i)
    ctx.close()
    for wp in progs.values():
        wp.close()


def test_build_refuses_what_is_not_a_template(rsv):
    """A template has to verify under the given configuration and inputs; anything else is an error code, not a program."""
    name = "level12-1.bin"
    proof = read_proof(name)
    for bad_proof, cfg, inputs, code in ((ob.tamper(proof, 5), fixture_cfg(name), _inputs(name), -5), (proof, fixture_cfg("level9-1.bin"), _inputs(name), -2),
                                         (proof, fixture_cfg(name), [(1, (1, 0, 0, 0))], -5), (proof[:4000], fixture_cfg(name), _inputs(name), -2)):
        with pytest.raises(rsv.RsvError) as e:
            rsv.WitnessProgram.build(bad_proof, cfg, inputs)
        assert e.value.code == code


def test_witness_of_a_ragged_large_batch(rsv):
    """n = 600 (three 256-lane blocks per instruction, the last one ragged; a tampered proof in the middle): EVERY row of
    both output layouts against the gadgets' vector."""
    name = "level12-1.bin"
    wp = rsv.WitnessProgram.build(read_proof(name), fixture_cfg(name), _inputs(name))
    want, _, _ = _oracle_variables(name)
    n, bad = 600, 311
    batch = [read_proof(name)] * n
    batch[bad] = ob.tamper(read_proof(name), 3)
    good = np.arange(n) != bad
    variables, accept, reason = rsv.witness(batch, wp, _inputs(name))[:3]
    assert accept.tolist() == good.astype(int).tolist() and reason[bad] != 0
    assert (variables[good] == want[None]).all()
    wp.close()


def test_witness_launch_forms_agree(rsv):
    """The level-per-launch form (its tail as a strip or in one launch) and the one-launch form (a workgroup per 1, 4, 64
    proofs), forced through RSV_OPT_WITNESS_SMALL_MAX / _SMALL_LOG / _WALK_LOG on batches of 3 and 70 proofs with a tampered one:
    the same rows as the gadgets'."""
    import torch
    name = "level12-1.bin"
    wp = rsv.WitnessProgram.build(read_proof(name), fixture_cfg(name), _inputs(name))
    want, _, _ = _oracle_variables(name)
    dev = torch.device("cuda:0")
    ctx = rsv.Context(0)
    for n, bad in ((3, 1), (70, 41)):
        batch = [read_proof(name)] * n
        batch[bad] = ob.tamper(read_proof(name), 5)
        blob, offsets = rsv.pack(batch)
        d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
        good = np.arange(n) != bad
        # (small_max, small_log, walk_log): level form with its tail as a strip / in one launch of 2, 8, 64 proofs per workgroup /
        # by the default rule; the whole program in one launch of 1, 4, 64 proofs per workgroup; the default rule
        for small_max, small_log, walk_log in ((1, 0, 1), (1, 0, 2), (1, 0, 4), (1, 0, 7), (1, 0, 0), (n + 1, 1, 0), (n + 1, 3, 0),
                                               (n + 1, 7, 0), (0, 0, 0)):
            ctx.set_option("witness_small_max", small_max)
            ctx.set_option("witness_small_log", small_log)
            ctx.set_option("witness_walk_log", walk_log)
            d_vars = torch.full((n, wp.n_vars, 4), -1, dtype=torch.int32, device=dev)
            d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
            ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=_inputs(name))
            ctx.synchronize()
            assert d_acc.cpu().numpy().tolist() == good.astype(int).tolist(), (n, small_max, small_log, walk_log)
            assert (d_vars.cpu().numpy().view(np.uint32)[good] == want[None]).all(), (n, small_max, small_log, walk_log)
        # the verifying pass's tree kernels in the lane form and in the row form (what batches this small take by themselves)
        for trees in ("paced", "row16", "auto"):
            ctx.set_option("tree_pace", trees)
            d_vars = torch.full((n, wp.n_vars, 4), -1, dtype=torch.int32, device=dev)
            d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
            ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=_inputs(name))
            ctx.synchronize()
            assert d_acc.cpu().numpy().tolist() == good.astype(int).tolist(), (n, trees)
            assert (d_vars.cpu().numpy().view(np.uint32)[good] == want[None]).all(), (n, trees)
    for opt, v in (("witness_small_max", (1 << 20) + 2), ("witness_small_log", 8), ("witness_small_max", -1), ("witness_walk_log", 8)):
        with pytest.raises(rsv.RsvError):
            ctx.set_option(opt, v)
    ctx.close()
    wp.close()


def test_witness_on_device_buffers_and_a_wrong_configuration(rsv):
    """Context.witness on tensors in HBM (n = 96 copies, two calls on one context: the second reuses the scratch), and the
    API errors: a configuration that is not the program's, a misaligned output."""
    import torch
    name = "level12-1.bin"
    built = rsv.WitnessProgram.build(read_proof(name), fixture_cfg(name), _inputs(name))
    prog = built.export()
    want, _, _ = _oracle_variables(name)
    wp = rsv.WitnessProgram(prog)  # the same program through rsv_witness_program_create, as a host that loads a file would
    built.close()
    n = 96
    blob, offsets = rsv.pack([read_proof(name)] * n)
    dev = torch.device("cuda:0")
    d_blob, d_off = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_vars = torch.zeros((n, prog.n_vars, 4), dtype=torch.int32, device=dev)
    d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx = rsv.Context(0)
    for _ in range(2):
        d_vars.zero_()
        ctx.witness(wp, d_blob, d_off, n, d_vars, d_acc, inputs=_inputs(name))
        ctx.synchronize()
        got = d_vars.cpu().numpy().view(np.uint32)
        assert d_acc.cpu().numpy().all() and all(np.array_equal(got[k], want) for k in (0, 1, n // 2, n - 1))
    assert wp.scratch_bytes(n) > n * prog.n_vars * 16
    # the other output layout: [variable][proof], as the level kernels write it (no transpose, no second copy)
    ctx.set_option("witness_layout", "by_variable")
    d_byvar = torch.zeros((prog.n_vars, n, 4), dtype=torch.int32, device=dev)
    ctx.witness(wp, d_blob, d_off, n, d_byvar, d_acc, inputs=_inputs(name))
    ctx.synchronize()
    byvar = d_byvar.cpu().numpy().view(np.uint32)
    assert np.array_equal(byvar[:, 0], want) and np.array_equal(byvar[:, n - 1], want) and np.array_equal(byvar[:, 17], want)
    ctx.set_option("witness_layout", "by_proof")
    other = rsv.WitnessProgram(prog)
    other.shape.n_queries = 9  # what cfg() reports to the call below
    with pytest.raises(rsv.RsvError) as e:
        ctx.witness(other, d_blob, d_off, n, d_vars, d_acc, inputs=_inputs(name))
    assert e.value.code == -2
    ctx.close()
    wp.close()
    other.close()
