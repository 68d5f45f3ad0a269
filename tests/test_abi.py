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
    assert lib.rsv_abi_version() == 6
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


def test_options_are_explicit_and_validated(rsv):
    """The library reads no environment variable: every knob goes through rsv_ctx_set_option (ctx = NULL: process
    default), unknown options and out-of-range values are refused — no device needed for any of this."""
    lib = rsv.lib
    assert lib.rsv_ctx_set_option(None, 999, 1) == -2          # RSV_E_SIZE: unknown option
    assert lib.rsv_ctx_set_option(None, 0, 1) == -2
    for name in ("transcript_form", "transcript_split", "oods_form", "qconst_form", "plan_form", "tree_cap", "overlap_trees",
                 "critical_chain", "device_order", "graph", "witness_layout", "cap_top", "flow_cap", "pair_order", "stage_times", "query_form"):
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], 3) == -5, name   # RSV_E_RANGE
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], -1) == -5, name
        for v in (2, 1, 0):
            assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], v) == 0, name
    for v, want in ((4, -5), (-1, -5), (3, 0), (2, 0), (1, 0), (0, 0)):  # four settings: auto, paced, unpaced, row form
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS["tree_pace"], v) == want, v
    for name, bad, good in (("ws_budget_mb", 0, 8192), ("perm_wg_per_cu", 9, 8), ("host_chunk_mb", 0, 256), ("host_threads", 65, 0),
                            ("debug_log", 2, 0), ("witness_small_max", (1 << 20) + 2, 0), ("witness_small_log", 8, 0), ("witness_walk_log", 8, 0)):
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], bad) == -5, name
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], good) == 0, name
    # no getenv anywhere in the product sources, no RSV_ variable in the library's strings
    import subprocess
    for root, _, files in os.walk(os.path.join(ROOT, "recursive-stwo_amd", "csrc")):
        for f in files:
            if f.endswith((".hpp", ".hip", ".inc")):
                assert "getenv" not in open(os.path.join(root, f)).read(), f
    syms = subprocess.run(["nm", "-D", "--undefined-only", rsv.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_poseidon_flow_count_is_host_arithmetic(rsv):
    """rsv_poseidon_flow_count needs no device; SURVEY App. C shapes: small_proof 3 481, recursive_proof_16_15 5 289."""
    from tests.conftest import fixture_cfg
    assert rsv.poseidon_flow_count(4, 8, fixture_cfg("small_proof.bin")) == 3481
    assert rsv.poseidon_flow_count(16, 15, fixture_cfg("recursive_proof_16_15.bin")) == 5289
    assert rsv.poseidon_flow_count(19, 18, fixture_cfg("level1-5.bin")) == 28095
    with pytest.raises(rsv.RsvError):
        rsv.poseidon_flow_count(0, 8, fixture_cfg("small_proof.bin"))


def test_rust_mirror_matches_header():
    """INTEGRATION.md §1 (the Rust `extern "C"` block of the rsv-sys crate) is generated from include/rsv.h by
    tools/gen_rust_ffi.py; no Rust compiler exists in the image, so this diff is what keeps the mirror from drifting:
    every declared function appears exactly once with the header's argument list, every struct field and constant too."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.function_names() == declared_functions()          # the generator's parser sees what this file's regex sees
    want, have = gen.generate(), gen.committed_block()
    assert have == want, "INTEGRATION.md §1 differs from include/rsv.h: run `python tools/gen_rust_ffi.py --write`"
    for name in declared_functions():
        assert have.count(f"pub fn {name}(") == 1, name
    # spot checks of the type mapping
    assert "pub fn rsv_verify_batch_host(ctx: *mut rsv_ctx, proofs: *const *const u8, lens: *const u64, n: usize," in have
    assert "pub fn rsv_last_stage_times(ctx: *mut rsv_ctx, names: *mut *const c_char, ms: *mut f32, cap: c_int) -> c_int;" in have
    assert "pub struct rsv_public_input { pub idx: u32, pub value: [u32; 4] }" in have
    assert "pub fn rsv_ctx_set_option(ctx: *mut rsv_ctx, option: c_int, value: i64) -> c_int;" in have
