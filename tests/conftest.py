import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["proofs"]


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
