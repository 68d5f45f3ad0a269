import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _install_native_backtrace():
    """A native stack trace on SIGSEGV & co. for the whole session (tests/native_backtrace.c): faulthandler alone names only
    the Python frame, which is all that round 3's one host crash left behind.  Best effort: no compiler, no handler."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "native_backtrace.c")
    lib = os.path.join(ROOT, "tests", "native_backtrace.so")
    try:
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-o", lib, src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        fd = os.dup(sys.__stderr__.fileno())  # (as pytest's own faulthandler plugin does: descriptor 2 is captured while tests run)
        return ctypes.CDLL(lib).rsv_test_install_native_backtrace(fd) == 0
    except (OSError, subprocess.CalledProcessError):
        return False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # behind pytest's own faulthandler plugin (which installed its handlers before conftest files are configured): ours
    # runs first on a fatal signal and then chains to it
    config._rsv_native_backtrace = _install_native_backtrace()


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["proofs"]


class Cfg:
    """PcsConfig literal of a fixture as written in the reference source (manifest.json cites file:line); both
    bindings accept any object with these four attributes."""

    def __init__(self, pow_bits, log_blowup_factor, log_last_layer_degree_bound, n_queries):
        self.pow_bits, self.log_blowup_factor = pow_bits, log_blowup_factor
        self.log_last_layer_degree_bound, self.n_queries = log_last_layer_degree_bound, n_queries

    def __repr__(self):
        return f"Cfg(pow={self.pow_bits}, blowup={self.log_blowup_factor}, last={self.log_last_layer_degree_bound}, nq={self.n_queries})"


_DERIVED = {"small_proof_composition.bin": "small_proof.bin", "small_proof_dup_query.bin": "small_proof.bin",
            "recursive_proof_16_15_composition.bin": "recursive_proof_16_15.bin", "level1-5_dup_query.bin": "level1-5.bin"}


def fixture_cfg(name):
    """The configuration the reference verifies fixture `name` under (tests/golden/manifest.json)."""
    name = _DERIVED.get(name, name)
    e = next(e for e in load_manifest() if e["file"] == name)
    return Cfg(e["pow_bits"], e["log_blowup_factor"], e["log_last_layer_degree_bound"], e["n_queries"])


def read_proof(name):
    with open(os.path.join(GOLDEN, "proofs", name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def manifest():
    return load_manifest()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def rsv():
    """The product binding (loads csrc/librsv_hip.so; raises if it is not built)."""
    import rsvload
    return rsvload.load_package()


class _Knobs:
    """Process-default tuning knobs of the product library (rsv_ctx_set_option with ctx = NULL), restored to
    "automatic" when the test ends.  The library never reads the environment."""

    DEFAULTS = {"ws_budget_mb": 8192, "perm_wg_per_cu": 24, "host_chunk_mb": 256}

    def __init__(self, rsv):
        self.rsv, self.touched = rsv, set()

    def set(self, name, value):
        self.touched.add(name)
        self.rsv.set_default_option(name, value)

    def reset(self):
        for name in self.touched:
            self.rsv.set_default_option(name, self.DEFAULTS.get(name, 0))
        self.touched.clear()


@pytest.fixture
def knobs(rsv):
    k = _Knobs(rsv)
    yield k
    k.reset()
