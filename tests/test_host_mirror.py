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
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "proofs")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
