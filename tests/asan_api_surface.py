"""TEST INFRASTRUCTURE: the library's no-device API surface, run by tests/test_host_logic_asan.py inside a process that
preloads the AddressSanitizer runtime and loads the host-ASan build of the library (make -C recursive-stwo_amd/csrc asan):
argument validation, options, the pure host entry points (flow count, shard ranges, bitmap assembly), the host-side program
check of rsv_witness_program_create, and every device entry point failing cleanly with RSV_E_DEVICE in this container.
Usage: LD_PRELOAD=<libclang_rt.asan> python tests/asan_api_surface.py <librsv_hip_asan.so>"""
import sys, ctypes, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload
rsv = rsvload.load_package(lib_path=sys.argv[1])
lib = rsv.lib
out = np.zeros(16, np.uint32)
assert lib.rsv_abi_version() == 6
# argument validation, options, pure host entry points: no device in this container, every device call must fail cleanly
assert lib.rsv_poseidon2_permute(None, out.ctypes.data_as(rsv._u32p), 1, 0) == -1
assert lib.rsv_ctx_create(0, None) == -1
h = ctypes.c_void_p()
assert lib.rsv_ctx_create(0, ctypes.byref(h)) == -3
for name in rsv.OPTIONS:
    lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], 10**12)
    lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], -1)
    lib.rsv_ctx_set_option(None, rsv.OPTIONS[name], 0)
cnt = ctypes.c_uint32()
for lp in range(0, 32):
    for nq in (0, 1, 16, 128, 129):
        lib.rsv_poseidon_flow_count(lp, 15, ctypes.byref(rsv.PcsConfig(20, 5, 8, nq)), ctypes.byref(cnt))
rng = np.random.default_rng(0)
for n, w in [(10, 3), (1000, 7), (65, 2), (1, 1), (31, 4), (4097, 8), (3, 8), (0, 1)]:
    for r in range(w):
        rsv.shard_range(n, r, w)
    sw = max(1, (rsv.shard_range(n, 0, w)[1] + 31) // 32)
    g = rng.integers(0, 2**32, (w, sw), dtype=np.uint64).astype(np.uint32)   # garbage above the slices' own bits
    if n:
        rsv.exchange_assemble(n, w, g)
# round 5: the byte-balanced plan, the planned assembly (unequal slice widths, garbage above every slice's own bits), the
# configuration check — pure host arithmetic under the sanitizers
for trial in range(200):
    n = int(rng.integers(0, 300))
    w = int(rng.integers(1, 20))
    lens = rng.integers(0, 1 << 20, n, dtype=np.uint64) if trial % 4 else np.zeros(n, np.uint64)
    lo, hi = rsv.shard_plan(lens, w)
    assert lo[0] == 0 and hi[-1] == n and all(lo[r + 1] == hi[r] and hi[r] >= lo[r] for r in range(w - 1))
    if n:
        sw = max(1, max((h - l + 31) // 32 for l, h in zip(lo, hi)))
        g = rng.integers(0, 2**32, (w, sw), dtype=np.uint64).astype(np.uint32)
        acc, bm = rsv.exchange_assemble(n, w, g, plan=(lo, hi))
        assert len(acc) == n and np.array_equal(np.unpackbits(bm.view(np.uint8), bitorder="little")[:n], acc)
lo_a, hi_a = (ctypes.c_size_t * 3)(0, 5, 4), (ctypes.c_size_t * 3)(5, 4, 9)          # not a plan: RSV_E_SIZE, nothing read beyond it
assert lib.rsv_exchange_assemble_plan(3, lo_a, hi_a, np.zeros(8, np.uint32).ctypes.data_as(rsv._u32p), np.zeros(16, np.uint8).ctypes.data_as(rsv._u8p), None) == -2
for c in ((20, 5, 8, 16), (31, 5, 8, 16), (20, 0, 8, 16), (20, 5, 17, 16), (20, 5, 8, 129), (2**32 - 1,) * 4):
    rsv.cfg_check(rsv.PcsConfig(*c))
for cfg in (rsv.PcsConfig(20, 5, 8, 129), rsv.PcsConfig(20, 17, 8, 16)):            # beyond the limits: RSV_E_SIZE before any device call
    try:
        rsv.verify_batch([b"\0" * 64], cfg)
        raise SystemExit("a configuration beyond the limits was accepted")
    except rsv.RsvError as e:
        assert e.code == -2, e.code
mh = ctypes.c_void_p()
assert lib.rsv_multi_create((ctypes.c_int * 2)(0, 1), 2, ctypes.byref(mh)) == -3 and not mh.value
assert lib.rsv_host_alloc(1 << 20, ctypes.byref(mh)) == -3
lib.rsv_host_free(None)
proof = open(os.path.join(ROOT, "tests", "golden", "proofs", "small_proof.bin"), "rb").read()
try:
    rsv.verify_batch([proof], rsv.PcsConfig(20, 5, 2, 16))
except rsv.RsvError as e:
    assert e.code == -3
try:
    rsv.WitnessProgram.build(proof, rsv.PcsConfig(20, 5, 2, 16), [(1, (1, 0, 0, 0))])
except rsv.RsvError as e:
    assert e.code == -3, e.code
# rsv_witness_program_create: the whole program is validated on the host before any device call
prog = np.zeros((4, 8), np.uint32); prog[:, 1] = np.arange(4)
lv = np.array([0, 4], np.uint32)
shape = rsv.WitnessShape(2, 3, 20, 5, 2, 16, 1, 10, 1)
ph = ctypes.c_void_p()
assert lib.rsv_witness_program_create(prog.ctypes.data_as(rsv._u32p), 4, lv.ctypes.data_as(rsv._u32p), 1, 4, ctypes.byref(shape), 0, ctypes.byref(ph)) == -3
prog[2, 0] = 99
assert lib.rsv_witness_program_create(prog.ctypes.data_as(rsv._u32p), 4, lv.ctypes.data_as(rsv._u32p), 1, 4, ctypes.byref(shape), 0, ctypes.byref(ph)) == -5
print("asan api ok")
