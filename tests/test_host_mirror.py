"""Builds and runs the C++ restatement of the reference's unit tests against the host mirror."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_mirror_test.cpp")
CSRC = os.path.join(ROOT, "recursive-stwo_amd", "csrc")


def build(tmp_path):
    exe = os.path.join(str(tmp_path), "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, SRC, "-L" + CSRC, "-lrsv_hip",
                           "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_host_mirror_compiles(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_host_mirror_runs_reference_tests(tmp_path):
    exe = build(tmp_path)
    # the witness program of the level10 shape, written by the Python side of the host layer and loaded by the C++ side
    import rsvload
    from tests.conftest import fixture_cfg, read_proof
    rsv = rsvload.load_package()
    prog = rsv.WitnessProgram.build(read_proof("level10-1.bin"), fixture_cfg("level10-1.bin")).export()
    path = os.path.join(str(tmp_path), "level10.rsvw")
    prog.save_raw(path)
    back = rsv.witness_program.Program.load_raw(path)
    assert back.n_vars == prog.n_vars and back.shape == prog.shape and (back.instr == prog.instr).all()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "proofs"), path], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout and f"witness: {prog.n_vars} variables per proof, accept = 1 1 0 0" in out.stdout


def _build_example(tmp_path, name):
    exe = os.path.join(str(tmp_path), name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "examples", name + ".cpp"),
                           "-L" + CSRC, "-lrsv_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_examples_compile(tmp_path):
    _build_example(tmp_path, "single_proof")
    _build_example(tmp_path, "multi_proofs")


@pytest.mark.gpu
def test_examples_run(tmp_path):
    """The C++ counterparts of the reference's examples/single-proof and examples/multi-proofs."""
    proofs = os.path.join(ROOT, "tests", "golden", "proofs")
    out = subprocess.run([_build_example(tmp_path, "single_proof"), os.path.join(proofs, "small_proof.bin")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "proof accepted" in out.stdout and "192 per-query Merkle paths" in out.stdout, out.stdout + out.stderr
    # the recursion circuit of examples/single-proof verifies small_proof.bin once: 3 481 Poseidon invocations, which pad
    # to the 2^15 Poseidon rows in the header of the proof it writes (recursive_proof_16_15.bin)
    assert "3481 invocations -> log_size_poseidon 15" in out.stdout, out.stdout
    # and its Plonk half: the 52 113 variables behind the 45 870 rows that pad to the 2^16 of that header
    assert "Plonk circuit: 52113 variables (the first witnesses: log sizes 4 / 8), 3481 flow entries with wires" in out.stdout, out.stdout
    files = [os.path.join(proofs, f) for f in ("level1-5.bin", "level7-1.bin", "level13-1.bin", "hybrid_hash.bin")]
    out = subprocess.run([_build_example(tmp_path, "multi_proofs")] + files, capture_output=True, text=True)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 1 and len(lines) == 4, out.stdout + out.stderr
    assert all("accepted" in l for l in lines[:3]) and "REJECTED (stage 1)" in lines[3]
