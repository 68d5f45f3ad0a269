"""The oracle's restatement of the recursion circuit (oracle/recursion_circuit: constraint system, gadgets, the five verifier
stages, witness program) against the reference's own fixtures — CPU only.

Each fixture of the reference's chain is the proof of the circuit that verifies the previous one
(examples/single-proof/src/main.rs, examples/multi-proofs/src/main.rs:173-295).  So the circuit restated here, run on
fixture K with the CPU oracle's hints, must BE what fixture K+1 proves:
  * its Plonk rows, padded to a power of two, and its Poseidon invocations, padded and laid out six rows each, give the
    two log sizes in K+1's header;
  * the interpolant of each of its columns — 10 preprocessed + 12 trace columns of the Plonk component (wires, op,
    multiplicities; the `variables` vector read through the wires), 40 + 48 of the Poseidon component (the PoseidonFlow)
    — evaluated at K+1's OODS point is the sampled value K+1 carries for that column: 110 QM31 equalities per pair.
tests/pin_recursion_circuit.py checks all 14 pairs and records them in tests/golden/recursion_circuit_pins.json together
with the one thing that has to be searched (the order in which the reference walked two HashSets, see oracle/recursion_circuit/verifier.py);
here a subset is re-checked from that file.  The same run also cross-checks the two restatements of the circuit's Poseidon
invocations: cs.flow must equal the C oracle's rsvo_poseidon_flow record for record.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.conftest import GOLDEN, load_manifest, read_proof

rc = pytest.importorskip("oracle.recursion_circuit")
from oracle.recursion_circuit import trace as T  # noqa: E402

with open(os.path.join(GOLDEN, "recursion_circuit_pins.json")) as f:
    PINS = {p["src"]: p for p in json.load(f)["pairs"]}
MAN = {e["file"]: e for e in load_manifest()}


def _inputs(name):
    return [(i, tuple(v)) for i, v in MAN[name]["inputs"]]


def _round_constants():
    ob.lib.rsvo_round_constants.restype = ctypes.POINTER(ctypes.c_uint32)
    r = [ob.lib.rsvo_round_constants(k) for k in range(3)]
    return ([[int(r[0][16 * a + i]) for i in range(16)] for a in range(4)], [int(r[1][i]) for i in range(14)],
            [[int(r[2][16 * a + i]) for i in range(16)] for a in range(4)])


def test_every_consecutive_fixture_pair_is_pinned():
    """The committed record covers the whole chain: 14 pairs, every one with 22/22 and 88/88 (the tool asserts that before
    it writes a pair), among them the two five-copy circuits."""
    pairs = list(PINS.values())
    assert len(pairs) == 14 and sum(p["multiplier"] == 5 for p in pairs) == 2
    for p in pairs:  # the header of dst is what the row / invocation counts predict
        w = np.frombuffer(read_proof(p["dst"])[:8], np.uint32)
        assert (p["plonk_rows"] - 1).bit_length() == int(w[0])
        assert ob.flow_log_size(p["poseidon_invocations"]) == int(w[1])


@pytest.mark.parametrize("src", ["small_proof.bin", "level6-1.bin", "level7-1.bin", "level9-1.bin", "level12-1.bin"])
def test_circuit_reproduces_the_next_fixtures_sampled_values(src):
    pin = PINS[src]
    inputs = _inputs(src)
    nxt = read_proof(pin["dst"])
    c, d, _ = rc.build_circuit(read_proof(src), ob, inputs, pin["multiplier"], [tuple(tuple(x) for x in o) for o in pin["shift_orders"]])
    assert c.num_plonk_rows() == pin["plonk_rows"] and len(c.flow) == pin["poseidon_invocations"]
    c.check_arithmetics()
    # the circuit's Poseidon invocations == the C oracle's restatement of them (oracle/rsv_oracle.c, rsvo_poseidon_flow)
    mine = np.array([list(e1[1]) + list(e2[1]) + list(e3[1]) + list(e4[1]) + [int(sw)] for (e1, e2, e3, e4, _a, sw) in c.flow], dtype=np.uint32)
    assert np.array_equal(mine, ob.poseidon_flow(read_proof(src), inputs))
    tr = ob.transcript_raw(nxt)
    oods = (tuple(int(x) for x in tr[20:24]), tuple(int(x) for x in tr[24:28]))
    want = rc.parse_proof(nxt).sampled_values
    lp, lq = (int(x) for x in np.frombuffer(nxt[:8], np.uint32))
    assert T.pad(c) == 1 << lp
    pe = T.PointEvaluator(lp, oods)
    pre, trace = T.plonk_columns(c)
    for k, name in enumerate(T.PREPROCESSED):
        assert pe.eval(pre[name]) == want[0][k][0], name
    for k in range(12):  # a_val, b_val, c_val: the `variables` vector read through the wires
        assert pe.eval(trace[k]) == want[1][k][0], k
    qpre, qtr = T.poseidon_columns(c.flow, _round_constants(), lq, padding_hash=([0] * 8,))
    pe = T.PointEvaluator(lq, oods)
    for k in range(40):
        assert pe.eval(qpre[k]) == want[0][10 + k][0], k
    for k in range(48):  # in / intermediate / out state of every round of every invocation
        assert pe.eval(qtr[k]) == want[1][12 + k][0], k


def test_point_evaluator_on_a_known_polynomial():
    """The evaluator itself: the interpolant of f(x, y) = 3 + 5x + 7y + 11xy given on the domain is f at any other point."""
    C = rc.C
    n = 6
    half = rc.gadgets.canonic_half_coset(n)
    hx, hy = T._coset_points(half.initial_index, half.step_size, n - 1)
    xs = list(hx) + list(hx)
    ys = list(hy) + [(-int(v)) % C.P for v in hy]
    nat = [(3 + 5 * int(x) + 7 * int(y) + 11 * int(x) * int(y)) % C.P for x, y in zip(xs, ys)]
    rev = T._bit_reverse_perm(n)
    col = [nat[int(r)] for r in rev]
    # a point of the circle over QM31: from a parameter t, as the OODS point is made (circle/src/lib.rs:204-219)
    t = (12345, 678, 91011, 1213)
    t2 = C.q_mul(t, t)
    inv = C.q_inv(C.q_add(t2, C.ONE4))
    pt = (C.q_mul(C.q_sub(C.ONE4, t2), inv), C.q_mul(C.q_add(t, t), inv))
    want = C.q_add(C.q_add(C.q_add((3, 0, 0, 0), C.q_scale(pt[0], 5)), C.q_scale(pt[1], 7)), C.q_scale(C.q_mul(pt[0], pt[1]), 11))
    assert T.PointEvaluator(n, pt).eval(col) == want


def _program_for(name, copies=1):
    c, d, _ = rc.build_circuit(read_proof(name), ob, _inputs(name), copies)
    return rc.program.extract(c, d, copies), c, d


def _product_program(prog):
    """The oracle's program in the product's container (what rsv.WitnessProgram takes, and what it saves and loads)."""
    import rsvload
    return rsvload.load_package().witness_program.Program(prog.instr, prog.level_offsets, prog.n_vars, prog.shape, prog.flow_wires)


def test_program_files_round_trip(tmp_path):
    import rsvload
    P = rsvload.load_package().witness_program.Program
    prog = _product_program(_program_for("level12-1.bin")[0])
    for save, load, name in ((prog.save, P.load, "p.npz"), (prog.save_raw, P.load_raw, "p.rsvw")):
        path = str(tmp_path / name)
        save(path)
        back = load(path)
        assert back.n_vars == prog.n_vars and back.shape == prog.shape
        assert np.array_equal(back.instr, prog.instr) and np.array_equal(back.level_offsets, prog.level_offsets)
    with open(str(tmp_path / "junk"), "wb") as f:
        f.write(b"\0" * 100)
    with pytest.raises(ValueError):
        P.load_raw(str(tmp_path / "junk"))


def test_program_depends_on_the_shape_only():
    """level10-1 and level11-1 are two different proofs of one shape: the programs extracted from them are identical, so a
    program built from one template serves every proof of its shape — also where the reference's own gate list follows the
    witness (the `op` of CirclePointM31Var::select, gadgets.pm_select)."""
    p10, _, _ = _program_for("level10-1.bin")
    p11, _, _ = _program_for("level11-1.bin")
    assert p10.n_vars == p11.n_vars and p10.shape == p11.shape
    assert np.array_equal(p10.level_offsets, p11.level_offsets) and np.array_equal(p10.instr, p11.instr)
    assert read_proof("level10-1.bin") != read_proof("level11-1.bin")


class _Sources:
    """What the GPU kernel reads, from the oracle's hints of one proof (for the host interpreter of a program)."""

    def __init__(self, proof, d, flow):
        self.w = np.frombuffer(proof, np.uint32)
        self.d, self._flow = d, flow

    def word(self, i): return int(self.w[i])
    def fri_commit(self, layer, half):
        c = self.d.first_layer_commitment if layer == 0 else self.d.inner_layer_commitments[layer - 1]
        return tuple(c[4 * half:4 * half + 4])
    def last_poly(self, k): return self.d.last_poly[k]
    def nonce(self, part):
        n = self.d.nonce
        return n & ((1 << 22) - 1) if part == 0 else (n >> (22 if part == 1 else 43)) & ((1 << 21) - 1)
    def flow(self, k, word): return tuple(int(x) for x in self._flow[k][word:word + 4])
    def trace_col(self, t, i, j): return int(self.d.raw_trace_cols[t, i, j])
    def fri_col(self, tree, i, word): return tuple(int(x) for x in self.d.raw_fri_cols[tree, i].reshape(-1)[word:word + 4])


@pytest.mark.parametrize("template,other,copies", [("level10-1.bin", "level11-1.bin", 1), ("level11-1.bin", "level10-1.bin", 2)])
def test_program_evaluates_to_the_variables_of_another_proof(template, other, copies):
    """Program from one proof, evaluated (host interpreter, Python integers) on ANOTHER proof's hints == the `variables`
    the gadgets compute when they run on that proof directly — every hint tag, every operand index, and the level order
    (an instruction may only read what earlier levels wrote), also with two copies of the verifier in one circuit."""
    prog, _, _ = _program_for(template, copies)
    lv = prog.level_offsets
    assert lv[0] == 0 and lv[-1] == prog.n_vars == len(prog.instr) and (np.diff(lv.astype(np.int64)) > 0).all()
    c, d, _ = rc.build_circuit(read_proof(other), ob, _inputs(other), copies)
    flow = ob.poseidon_flow(read_proof(other), _inputs(other))
    got = rc.program.interpret(prog, _Sources(read_proof(other), d, flow))
    assert got == [tuple(v) for v in c.variables]
    # level discipline: walk the sorted list, every operand already written
    written = np.zeros(prog.n_vars, bool)
    P_ = rc.program
    for l in range(len(lv) - 1):
        blk = prog.instr[lv[l]:lv[l + 1]]
        two, one = np.isin(blk[:, 0], (P_.ADD, P_.MUL)), np.isin(blk[:, 0], (P_.MULC, P_.COPY, P_.INV, P_.INV0, P_.QINV, P_.CINV, P_.COORD, P_.BIT))
        assert written[blk[two | one, 2]].all() and written[blk[two, 3]].all()
        written[blk[:, 1]] = True
    assert written.all()


def test_program_create_rejects_what_the_device_could_not_index():
    """rsv_witness_program_create checks the program before anything touches a device: RSV_E_RANGE for an operand of a
    later level, an index beyond a hint buffer, an unknown op; on this GPU-less box a valid program gets as far as
    RSV_E_DEVICE."""
    import rsvload
    rsv = rsvload.load_package()
    prog, _, _ = _program_for("level12-1.bin")

    def create(instr, levels=None):
        try:
            rsv.WitnessProgram(rsv.witness_program.Program(instr, prog.level_offsets if levels is None else levels, prog.n_vars, prog.shape,
                                                           prog.flow_wires))
        except rsv.RsvError as e:
            return e.code
        return 0

    have_gpu = rsv.device_count() > 0
    assert create(prog.instr) == (0 if have_gpu else -3)
    P_ = rc.program
    k = int(np.nonzero(prog.instr[:, 0] == P_.ADD)[0][-1])
    for col, val in ((2, prog.instr[-1, 1]), (0, 99), (1, prog.n_vars)):
        bad = prog.instr.copy()
        bad[k, col] = val
        assert create(bad) == -5
    for op, col, val in ((P_.FLOW, 4, prog.shape["flow_count"]), (P_.TRACE_COL, 6, 64), (P_.FRI_COL, 4, prog.shape["n_inner"] + 1),
                         (P_.WORD4, 4, 895), (P_.BIT, 4, 31), (P_.LAST_POLY, 4, 1 << prog.shape["log_last"])):
        bad = prog.instr.copy()
        bad[int(np.nonzero(bad[:, 0] == op)[0][0]), col] = val
        assert create(bad) == -5, P_.OP_NAMES[op]
    dup = prog.instr.copy()
    dup[k, 1] = dup[k - 1, 1]
    assert create(dup) == -5
    assert create(prog.instr, prog.level_offsets[:-1]) == -2
