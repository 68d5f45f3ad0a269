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
                 "critical_chain", "device_order", "graph", "witness_layout", "cap_top", "flow_cap", "pair_order", "stage_times", "query_form", "cap_mid", "oods_early", "tree_order"):
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], 3) == -5, name   # RSV_E_RANGE
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], -1) == -5, name
        for v in (2, 1, 0):
            assert lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], v) == 0, name
    for v, want in ((4, -5), (-1, -5), (3, 0), (2, 0), (1, 0), (0, 0)):  # four settings: auto, paced, unpaced, row form
        assert lib.rsv_ctx_set_option(None, rsv.OPTIONS["tree_pace"], v) == want, v
    for name, bad, good in (("ws_budget_mb", 0, 8192), ("perm_wg_per_cu", 33, 24), ("host_chunk_mb", 0, 256), ("host_threads", 65, 0),
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


def test_shape_limits_are_api_not_parse_errors(rsv):
    """include/rsv.h names the library's shape limits (RSV_MAX_*): a CONFIGURATION beyond them is RSV_E_SIZE from every entry
    point that takes an rsv_cfg_set — before any device work, so it shows without a GPU — never n proofs rejected as
    RSV_R_PARSE.  rsv_cfg_check tells in advance."""
    ok = rsv.PcsConfig(20, 5, 8, 16)
    assert rsv.cfg_check(ok) and rsv.cfg_check(rsv.PcsConfig(30, 16, 16, 128)) and rsv.cfg_check(rsv.PcsConfig(0, 1, 0, 1))
    assert rsv.lib.rsv_cfg_check(None) == -1
    beyond = [rsv.PcsConfig(20, 5, 8, 129), rsv.PcsConfig(20, 5, 8, 0), rsv.PcsConfig(20, 17, 8, 16), rsv.PcsConfig(20, 0, 8, 16),
              rsv.PcsConfig(20, 5, 17, 16), rsv.PcsConfig(31, 5, 8, 16)]
    proof = b"\0" * 64
    for cfg in beyond:
        assert not rsv.cfg_check(cfg)
        for call in (lambda: rsv.verify_batch([proof], cfg),                       # rsv_verify_batch
                     lambda: rsv.verify_batch([proof, proof], [ok, cfg]),          # one bad configuration in a set of two
                     lambda: rsv.verify_batch([], cfg)):                           # even for an empty batch
            with pytest.raises(rsv.RsvError) as e:
                call()
            assert e.value.code == -2, (rsv._cfg_key(cfg), e.value.code)          # RSV_E_SIZE, not RSV_E_DEVICE (-3)
    # the limits in the header are the ones the kernels are compiled with
    hdr = open(os.path.join(ROOT, "include", "rsv.h")).read()
    lay = open(os.path.join(ROOT, "recursive-stwo_amd", "csrc", "layout.hpp")).read()
    import re
    for macro, const in (("RSV_MAX_QUERIES", "MAXQ"), ("RSV_MAX_LOG_SIZE", "MAX_LOG"), ("RSV_MAX_FRI_INNER", "MAX_INNER")):
        assert re.search(rf"#define {macro} (\d+)", hdr).group(1) == re.search(rf"constexpr int {const} = (\d+);", lay).group(1)


def test_shard_plan_balances_the_level_ordered_chain(rsv):
    """rsv_shard_plan: contiguous cuts balanced by bytes.  The reference's own job arrives ordered by level
    (examples/multi-proofs/src/main.rs:198-295: level1-5 ... level13-1, 435 KB down to 76 KB): 13 x 4 096 proofs in level
    order over 8 ranks — cut by count rank 0 carries 4.5 x the bytes of rank 7, cut by the plan every rank the same within
    1.1 (missing 3).  No device needed."""
    from tests.conftest import read_proof
    names = ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin", "level7-1.bin",
             "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin", "level13-1.bin"]
    lens = np.repeat(np.array([len(read_proof(nm)) for nm in names], dtype=np.uint64), 4096)
    n, world = len(lens), 8
    pre = np.concatenate([[0], np.cumsum(lens)])
    by_count = [rsv.shard_range(n, r, world) for r in range(world)]
    b = [int(pre[hi] - pre[lo]) for lo, hi in by_count]
    assert max(b) / min(b) > 4.0                                                   # what the plan is for
    lo, hi = rsv.shard_plan(lens, world)
    assert lo[0] == 0 and hi[-1] == n and all(lo[r + 1] == hi[r] for r in range(world - 1))
    b = [int(pre[h] - pre[l]) for l, h in zip(lo, hi)]
    assert max(b) / min(b) <= 1.1 and max(b) - min(b) <= 2 * int(lens.max())
    # a uniform job: the plan is rsv_shard_range up to one proof per cut
    lo, hi = rsv.shard_plan(np.full(1000, 108896, np.uint64), 7)
    assert all(abs(lo[r] - rsv.shard_range(1000, r, 7)[0]) <= 1 for r in range(7))
    # degenerate jobs: more ranks than proofs (empty shards), an empty job, one giant proof, zero-length proofs
    for ln, w in (([5, 5, 5], 8), ([], 4), ([1, 1, 10 ** 9, 1], 3), ([0, 0, 0, 0], 2)):
        lo, hi = rsv.shard_plan(np.array(ln, np.uint64), w)
        assert lo[0] == 0 and hi[-1] == len(ln) and all(lo[r + 1] == hi[r] and hi[r] >= lo[r] for r in range(w - 1))
    assert rsv.lib.rsv_shard_plan(None, 0, 0, None, None) == -1
    lo_a, hi_a = (ctypes.c_size_t * 2)(), (ctypes.c_size_t * 2)()
    assert rsv.lib.rsv_shard_plan(None, 0, 5000, lo_a, hi_a) == -2                 # world beyond 4 096
    # the planned exchange's host half: slices of unequal width, garbage above a shard's own bits is masked
    lo, hi = [0, 40, 45], [40, 45, 110]
    acc = (np.arange(110) % 3 == 0).astype(np.uint8)
    sw = (65 + 31) // 32
    gathered = np.full((3, sw), 0xFFFFFFFF, np.uint32)
    for r in range(3):
        bits = np.zeros(sw * 32, np.uint8)
        bits[: hi[r] - lo[r]] = acc[lo[r]:hi[r]]
        bits[hi[r] - lo[r]:] = 1                                                    # stale words a careless producer left
        gathered[r] = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
    a2, bm = rsv.exchange_assemble(110, 3, gathered, plan=(lo, hi))
    assert a2.tolist() == acc.tolist() and np.array_equal(np.unpackbits(bm.view(np.uint8), bitorder="little")[:110], acc)
    with pytest.raises(rsv.RsvError):
        rsv.exchange_assemble(110, 3, gathered, plan=([0, 41, 45], hi))             # not contiguous


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
