"""CPU checks of the drop-in boundary: the product library loads, exports every symbol declared in
include/rsv.h, and — with no GPU present — fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rsv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    fns = declared_functions()
    for must in ("rsv_poseidon2_permute", "rsv_merkle_hash_node", "rsv_verify_batch", "rsv_verify_batch_dev",
                 "rsv_transcript", "rsv_accept_bitmap_dev"):
        assert must in fns


def test_library_exports_every_declared_symbol(rsv):
    lib = ctypes.CDLL(rsv.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in rsv.h but not exported by librsv_hip.so"
    assert lib.rsv_abi_version() == 3
    assert sorted(rsv.EXPORTS) == declared_functions()


def test_product_does_not_link_the_oracle(rsv):
    # the oracle is test infrastructure; the shipped library must not depend on it
    import subprocess
    out = subprocess.run(["ldd", rsv.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", rsv.LIB_PATH], capture_output=True, text=True).stdout
    assert "rsvo_" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "recursive-stwo_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".inc", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(root, f)).read().lower(), f


def test_no_cpu_fallback_without_gpu(rsv):
    if rsv.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(rsv.RsvError) as e:
        rsv.poseidon2_permute(np.arange(16, dtype=np.uint32))
    assert e.value.code == -3  # RSV_E_DEVICE
    with pytest.raises(rsv.RsvError):
        rsv.verify_batch([b"\0" * 64], rsv.PcsConfig(20, 5, 8, 16))
    with pytest.raises(rsv.RsvError):
        rsv.Context(0)


def test_argument_validation(rsv):
    # API misuse is reported as a status code before any device work
    out = np.zeros(16, np.uint32)
    assert rsv.lib.rsv_poseidon2_permute(None, out.ctypes.data_as(rsv._u32p), 1, 0) == -1
    assert rsv.lib.rsv_merkle_hash_node(None, None, None, 0, out.ctypes.data_as(rsv._u32p), 1, 0) == -2
    assert rsv.lib.rsv_transcript(None, 0, out.ctypes.data_as(rsv._u32p), 16, 0) == -1
    assert rsv.lib.rsv_ctx_create(0, None) == -1
