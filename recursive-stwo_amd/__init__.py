"""recursive-stwo_amd — host-side binding of the MI355X-native batch verifier.

This package is plumbing only: it loads ``csrc/librsv_hip.so`` (the C-ABI of
``include/rsv.h``) with ctypes and exposes the same entry points to Python,
either on host ``bytes``/numpy arrays (copied to the device by the library) or
on torch tensors that already live in HBM (``Context``).  There is no CPU
implementation here: if the library is missing, or no HIP device is usable,
every call raises.

Reference items mirrored (values only; see include/rsv.h for file:line):
  poseidon2_permute        primitives/poseidon31/src/implementation.rs:108-149
  half_permute             Poseidon2HalfVar::permute, primitives/poseidon31/src/lib.rs:282-423
  poseidon2_emulated       poseidon_permute_emulated (gate values), primitives/poseidon31/src/emulated.rs:80-221
  hash_node                Poseidon31MerkleHasherVar, primitives/merkle/src/lib.rs:9-181
  merkle_path_root         SinglePathMerkleProofVar::verify, components/recursive/data_structures/src/lib.rs:315-354
  transcript               FiatShamirResults::compute, components/recursive/fiat_shamir/src/lib.rs:31-176
  verify_batch             FiatShamirResults/CompositionCheck/AnswerResults/FoldingResults::compute
"""
from __future__ import annotations

import ctypes
import weakref
import os
from typing import Iterable, Optional, Sequence

import numpy as np

P = 0x7FFFFFFF
_HERE = os.path.dirname(os.path.abspath(__file__))
# (rsvload.load_package(lib_path=...) names a diagnostic build explicitly; nothing here reads the environment)
LIB_PATH = globals().get("_LIB_PATH_OVERRIDE") or os.path.join(_HERE, "csrc", "librsv_hip.so")

REASONS = ["ok", "parse", "pow", "logup", "composition", "dup_query", "merkle_t0", "merkle_t1",
           "merkle_t2", "merkle_t3", "fri_first", "fri_inner", "fri_last"]

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


class RsvError(RuntimeError):
    def __init__(self, code: int, what: str):
        names = {-1: "RSV_E_NULL", -2: "RSV_E_SIZE", -3: "RSV_E_DEVICE", -4: "RSV_E_CAP", -5: "RSV_E_RANGE", -6: "RSV_E_UNAVAILABLE"}
        super().__init__(f"{what}: {names.get(code, code)}")
        self.code = code


class PcsConfig(ctypes.Structure):
    """stwo PcsConfig{pow_bits, FriConfig::new(log_last_layer_degree_bound, log_blowup_factor, n_queries)}."""
    _fields_ = [("pow_bits", ctypes.c_uint32), ("log_blowup_factor", ctypes.c_uint32),
                ("log_last_layer_degree_bound", ctypes.c_uint32), ("n_queries", ctypes.c_uint32)]


class CfgSet(ctypes.Structure):  # rsv_cfg_set
    _fields_ = [("cfgs", ctypes.POINTER(PcsConfig)), ("n_cfgs", ctypes.c_uint32), ("cfg_of", ctypes.c_void_p)]


MAX_CFGS = 16  # RSV_MAX_CFGS


class PreparedCfg:
    """An rsv_cfg_set ready to pass by reference: keeps the config array (and the cfg_of buffer: a numpy array for
    the host entry points, a torch tensor in HBM for Context methods) alive."""

    def __init__(self, cfgs: Sequence[PcsConfig], cfg_of=None):
        if not 1 <= len(cfgs) <= MAX_CFGS:
            raise ValueError(f"1..{MAX_CFGS} configurations")
        self.arr = (PcsConfig * len(cfgs))(*[PcsConfig(c.pow_bits, c.log_blowup_factor, c.log_last_layer_degree_bound,
                                                       c.n_queries) for c in cfgs])
        self.cfg_of = cfg_of
        if cfg_of is None:
            ptr = None
        elif isinstance(cfg_of, np.ndarray):
            ptr = cfg_of.ctypes.data
        else:
            ptr = cfg_of.data_ptr()  # torch tensor
        self.struct = CfgSet(ctypes.cast(self.arr, ctypes.POINTER(PcsConfig)), len(cfgs), ptr)

    def ref(self):
        return ctypes.byref(self.struct)


def _cfg_key(c) -> tuple:
    return (int(c.pow_bits), int(c.log_blowup_factor), int(c.log_last_layer_degree_bound), int(c.n_queries))


def prepare_cfg(cfg, n: int, to_device=None) -> PreparedCfg:
    """cfg: one PcsConfig (every proof must carry it), or a sequence of n PcsConfig (one per proof; deduplicated into
    at most MAX_CFGS distinct configurations + a per-proof index), or a PreparedCfg.  The configuration is REQUIRED:
    the verifier never trusts the words serialized in a proof (include/rsv.h, rsv_cfg_set).
    to_device: callable numpy uint8 array -> device tensor (Context methods), None for the host entry points."""
    if isinstance(cfg, PreparedCfg):
        return cfg
    if cfg is None:
        raise TypeError("a PcsConfig (or one per proof) is required: the verifier does not trust the configuration "
                        "words serialized in a proof")
    if isinstance(cfg, PcsConfig) or hasattr(cfg, "pow_bits"):
        return PreparedCfg([cfg])
    per_proof = list(cfg)
    if len(per_proof) != n:
        raise ValueError("one configuration per proof expected")
    table, index = [], {}
    cfg_of = np.zeros(n, np.uint8)
    for i, c in enumerate(per_proof):
        k = _cfg_key(c)
        if k not in index:
            index[k] = len(table)
            table.append(c)
        cfg_of[i] = index[k]
    if len(table) <= 1:
        return PreparedCfg(table or [PcsConfig(0, 0, 0, 0)])
    return PreparedCfg(table, to_device(cfg_of) if to_device else cfg_of)


class PublicInput(ctypes.Structure):
    _fields_ = [("idx", ctypes.c_uint32), ("value", ctypes.c_uint32 * 4)]


class HintsOut(ctypes.Structure):  # rsv_hints_out
    _fields_ = [("n_queries", ctypes.c_uint32), ("max_log", ctypes.c_uint32), ("n_inner", ctypes.c_uint32),
                ("d_transcript", ctypes.c_void_p), ("d_trace_sib", ctypes.c_void_p), ("d_trace_pos", ctypes.c_void_p),
                ("d_trace_cols", ctypes.c_void_p), ("d_fri_sib", ctypes.c_void_p), ("d_fri_cols", ctypes.c_void_p),
                ("d_fri_folded", ctypes.c_void_p), ("d_query_values", ctypes.c_void_p),
                ("d_flow", ctypes.c_void_p), ("d_flow_swap", ctypes.c_void_p), ("d_flow_count", ctypes.c_void_p),
                ("flow_stride", ctypes.c_uint32), ("d_accept_bitmap", ctypes.c_void_p), ("d_accept_count", ctypes.c_void_p)]


class WitnessShape(ctypes.Structure):  # rsv_witness_shape
    _fields_ = [(k, ctypes.c_uint32) for k in ("log_size_plonk", "log_size_poseidon", "pow_bits", "log_blowup", "log_last", "n_queries",
                                               "n_inner", "flow_count", "copies")]


class Shard(ctypes.Structure):  # rsv_shard
    _fields_ = [("d_blob", ctypes.c_void_p), ("d_offsets", ctypes.c_void_p), ("n", ctypes.c_size_t), ("d_cfg_of", ctypes.c_void_p),
                ("d_accept", ctypes.c_void_p), ("d_reason", ctypes.c_void_p)]


TRANSCRIPT_WORDS = 284  # RSV_TRANSCRIPT_WORDS
EXCHANGE_ID_BYTES = 128  # RSV_EXCHANGE_ID_BYTES


def _load() -> ctypes.CDLL:
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  Import torch first so that
    # the process holds ONE runtime and the tensors torch allocates are visible to this library's stream.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    sz = ctypes.c_size_t
    vp = ctypes.c_void_p
    sig = {
        "rsv_abi_version": (ctypes.c_int, []),
        "rsv_device_count": (ctypes.c_int, []),
        "rsv_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
        "rsv_ctx_destroy": (None, [vp]),
        "rsv_ctx_synchronize": (ctypes.c_int, [vp]),
        "rsv_ctx_stream": (vp, [vp]),
        "rsv_ctx_set_option": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_longlong]),
        "rsv_poseidon2_permute": (ctypes.c_int, [_u32p, _u32p, sz, ctypes.c_int]),
        "rsv_poseidon2_permute_dev": (ctypes.c_int, [vp, vp, vp, sz, vp]),
        "rsv_ctx_wait_stream": (ctypes.c_int, [vp, vp]),
        "rsv_stream_wait_ctx": (ctypes.c_int, [vp, vp]),
        "rsv_poseidon2_half_permute": (ctypes.c_int, [_u32p, _u32p, _u8p, _u32p, _u32p, sz, ctypes.c_int]),
        "rsv_poseidon2_emulated": (ctypes.c_int, [_u32p, _u32p, _u8p, _u32p, sz, ctypes.c_int]),
        "rsv_poseidon2_emulated_dev": (ctypes.c_int, [vp, vp, vp, vp, vp, sz, vp]),
        "rsv_merkle_hash_node": (ctypes.c_int, [_u32p, _u32p, _u32p, sz, _u32p, sz, ctypes.c_int]),
        "rsv_merkle_path_root": (ctypes.c_int, [_u32p, _u32p, _u32p, _u32p, ctypes.c_uint32, _u32p, sz, ctypes.c_int]),
        "rsv_transcript": (ctypes.c_int, [_u8p, sz, _u32p, sz, ctypes.c_int]),
        "rsv_verify_batch": (ctypes.c_int, [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput),
                                            sz, _u8p, _u8p, ctypes.c_int]),
        "rsv_verify_batch_dev": (ctypes.c_int, [vp, vp, vp, sz, ctypes.POINTER(CfgSet),
                                                ctypes.POINTER(PublicInput), sz, vp, vp]),
        "rsv_accept_bitmap_dev": (ctypes.c_int, [vp, vp, sz, vp, vp]),
        "rsv_trace_paths_dev": (ctypes.c_int, [vp, vp, vp, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.c_uint32,
                                               ctypes.c_uint32, vp, vp, vp, vp]),
        "rsv_verify_hints_dev": (ctypes.c_int, [vp, vp, vp, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.POINTER(HintsOut), vp, vp]),
        "rsv_field_op": (ctypes.c_int, [ctypes.c_int, _u32p, _u32p, _u32p, sz, ctypes.c_int]),
        "rsv_domain_points": (ctypes.c_int, [ctypes.c_uint32, _u32p, _u32p, sz, ctypes.c_int]),
        "rsv_line_eval": (ctypes.c_int, [_u32p, ctypes.c_uint32, _u32p, _u32p, sz, ctypes.c_int]),
        "rsv_oods_eval": (ctypes.c_int, [_u32p, _u32p, _u32p, sz, ctypes.c_int]),
        "rsv_last_layer_check": (ctypes.c_int, [_u32p, ctypes.c_uint32, _u32p, _u32p, _u8p, sz, ctypes.c_int]),
        "rsv_verify_batch_host": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_void_p), _u64p, sz, ctypes.POINTER(CfgSet),
                                                 ctypes.POINTER(PublicInput), sz, _u8p, _u8p]),
        "rsv_verify_hints": (ctypes.c_int, [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.POINTER(HintsOut), _u8p, _u8p,
                                            ctypes.c_int]),
        "rsv_transcript_batch": (ctypes.c_int, [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), _u32p, ctypes.c_int]),
        "rsv_poseidon_flow_count": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(PcsConfig), _u32p]),
        "rsv_fri_paths_dev": (ctypes.c_int, [vp, vp, vp, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.c_uint32,
                                             ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp, vp]),
        "rsv_fri_paths": (ctypes.c_int, [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, _u32p, _u32p, _u8p, _u8p, ctypes.c_int]),
        "rsv_trace_paths": (ctypes.c_int, [_u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, ctypes.c_uint32,
                                           ctypes.c_uint32, _u32p, _u32p, _u8p, _u8p, ctypes.c_int]),
        "rsv_last_stage_times": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                                ctypes.c_int]),
        "rsv_witness_program_create": (ctypes.c_int, [_u32p, sz, _u32p, sz, ctypes.c_uint32, ctypes.POINTER(WitnessShape), ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_void_p)]),
        "rsv_witness_program_destroy": (None, [vp]),
        "rsv_witness_program_build": (ctypes.c_int, [_u8p, sz, ctypes.POINTER(PcsConfig), ctypes.POINTER(PublicInput), sz, ctypes.c_uint32,
                                                     _u8p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
        "rsv_witness_program_info": (ctypes.c_int, [vp, _u32p, _u32p, ctypes.POINTER(WitnessShape)]),
        "rsv_witness_program_export": (ctypes.c_int, [vp, _u32p, _u32p, _u32p]),
        "rsv_witness_program_gates": (ctypes.c_int, [vp, _u32p, _u32p, _u32p, _u32p]),
        "rsv_witness_scratch_bytes": (ctypes.c_int, [vp, sz, ctypes.POINTER(sz)]),
        "rsv_witness_eval_dev": (ctypes.c_int, [vp, vp, vp, vp, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, vp, vp, vp, vp, vp]),
        "rsv_witness_eval": (ctypes.c_int, [vp, _u8p, _u64p, sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz, _u32p, _u32p, _u8p,
                                            _u8p, _u8p, ctypes.c_int]),
        "rsv_host_alloc": (ctypes.c_int, [sz, ctypes.POINTER(vp)]),
        "rsv_host_free": (None, [vp]),
        "rsv_shard_range": (None, [sz, sz, sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]),
        "rsv_shard_plan": (ctypes.c_int, [_u64p, sz, sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]),
        "rsv_cfg_check": (ctypes.c_int, [ctypes.POINTER(PcsConfig)]),
        "rsv_exchange_create_plan": (ctypes.c_int, [vp, _u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(vp)]),
        "rsv_exchange_assemble_plan": (ctypes.c_int, [sz, ctypes.POINTER(sz), ctypes.POINTER(sz), _u32p, _u8p, _u32p]),
        "rsv_multi_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), sz, ctypes.POINTER(vp)]),
        "rsv_multi_destroy": (None, [vp]),
        "rsv_multi_size": (sz, [vp]),
        "rsv_multi_ctx": (vp, [vp, sz]),
        "rsv_multi_verify_batch_host": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_void_p), _u64p, sz, ctypes.POINTER(CfgSet),
                                                       ctypes.POINTER(PublicInput), sz, _u8p, _u8p, _u32p, _u64p]),
        "rsv_multi_verify_batch_dev": (ctypes.c_int, [vp, ctypes.POINTER(Shard), sz, ctypes.POINTER(CfgSet), ctypes.POINTER(PublicInput), sz,
                                                      _u32p, _u64p]),
        "rsv_exchange_available": (ctypes.c_int, []),
        "rsv_exchange_rccl_version": (ctypes.c_int, []),
        "rsv_exchange_unique_id": (ctypes.c_int, [_u8p]),
        "rsv_exchange_create": (ctypes.c_int, [vp, _u8p, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(vp)]),
        "rsv_exchange_destroy": (None, [vp]),
        "rsv_exchange_layout": (ctypes.c_int, [vp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]),
        "rsv_exchange_run": (ctypes.c_int, [vp, vp, vp, vp]),
        "rsv_exchange_assemble": (ctypes.c_int, [sz, sz, _u32p, _u8p, _u32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what rsv.h declares
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
EXPORTS = ["rsv_abi_version", "rsv_device_count", "rsv_ctx_create", "rsv_ctx_destroy", "rsv_ctx_synchronize",
           "rsv_ctx_stream", "rsv_ctx_set_option", "rsv_ctx_wait_stream", "rsv_stream_wait_ctx", "rsv_poseidon2_permute", "rsv_poseidon2_permute_dev", "rsv_poseidon2_half_permute",
           "rsv_poseidon2_emulated", "rsv_poseidon2_emulated_dev",
           "rsv_merkle_hash_node", "rsv_merkle_path_root", "rsv_transcript", "rsv_verify_batch",
           "rsv_verify_batch_dev", "rsv_accept_bitmap_dev", "rsv_last_stage_times", "rsv_trace_paths_dev",
           "rsv_trace_paths", "rsv_fri_paths_dev", "rsv_fri_paths", "rsv_verify_hints_dev", "rsv_verify_hints", "rsv_verify_batch_host", "rsv_field_op", "rsv_domain_points",
           "rsv_line_eval", "rsv_oods_eval", "rsv_last_layer_check",
           "rsv_transcript_batch", "rsv_poseidon_flow_count", "rsv_witness_program_create", "rsv_witness_program_destroy",
           "rsv_witness_program_build", "rsv_witness_program_info", "rsv_witness_program_export", "rsv_witness_program_gates",
           "rsv_witness_scratch_bytes", "rsv_witness_eval_dev", "rsv_witness_eval",
           "rsv_host_alloc", "rsv_host_free", "rsv_shard_range", "rsv_multi_create", "rsv_multi_destroy", "rsv_multi_size", "rsv_multi_ctx", "rsv_multi_verify_batch_host",
           "rsv_multi_verify_batch_dev", "rsv_exchange_available", "rsv_exchange_rccl_version", "rsv_exchange_unique_id",
           "rsv_exchange_create", "rsv_exchange_destroy", "rsv_exchange_layout", "rsv_exchange_run", "rsv_exchange_assemble",
           "rsv_shard_plan", "rsv_cfg_check", "rsv_exchange_create_plan", "rsv_exchange_assemble_plan"]


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RsvError(rc, what)


def _u32(a, shape=None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None:
        arr = arr.reshape(shape)
    return arr


def device_count() -> int:
    return lib.rsv_device_count()


#: rsv_option (include/rsv.h) by name, and the named values of the three-way knobs (0 is always "automatic")
OPTIONS = {"transcript_form": 1, "transcript_split": 2, "oods_form": 3, "qconst_form": 4, "plan_form": 5, "tree_cap": 6,
           "overlap_trees": 7, "ws_budget_mb": 8, "perm_wg_per_cu": 9, "host_chunk_mb": 10, "host_threads": 11, "debug_log": 12,
           "critical_chain": 13, "device_order": 14, "graph": 15,
           "witness_layout": 16, "witness_small_max": 17, "witness_small_log": 18, "cap_top": 19, "witness_walk_log": 20, "flow_cap": 21, "pair_order": 22, "tree_pace": 23, "stage_times": 24, "query_form": 25, "cap_mid": 26, "perm_form": 27, "oods_early": 28, "tree_order": 29}
OPTION_VALUES = {"auto": 0, "row": 1, "lane": 2, "paced": 1, "unpaced": 2, "row16": 3, "whole": 1, "split": 2, "parallel": 1, "serial": 2, "on": 1, "off": 2, "device": 1, "host": 2,
                 "by_proof": 1, "by_variable": 2}


def _set_option(handle, name: str, value) -> None:
    v = OPTION_VALUES[value] if isinstance(value, str) else int(value)
    _check(lib.rsv_ctx_set_option(handle, OPTIONS[name], v), f"rsv_ctx_set_option({name}={value})")


def set_default_option(name: str, value) -> None:
    """Process default of a tuning / diagnostic knob (rsv_ctx_set_option with ctx = NULL): inherited by every context
    created afterwards, including the ones the host-pointer entry points create for themselves.  The library never reads
    the environment."""
    _set_option(None, name, value)


def make_inputs(inputs: Iterable) -> "ctypes.Array[PublicInput]":
    """[(idx, (a0, a1, b0, b1)), ...] -> PublicInput array (e.g. (1, 1), (2, i), (3, u))."""
    items = list(inputs)
    arr = (PublicInput * max(len(items), 1))()
    for k, (idx, val) in enumerate(items):
        arr[k].idx = idx
        for t in range(4):
            arr[k].value[t] = int(val[t])
    return arr


#: public inputs used by examples/multi-proofs/src/main.rs:49-57
STANDARD_INPUTS = [(1, (1, 0, 0, 0)), (2, (0, 1, 0, 0)), (3, (0, 0, 1, 0))]


def poseidon2_permute(states, device: int = 0) -> np.ndarray:
    s = _u32(states).reshape(-1, 16)
    out = np.empty_like(s)
    _check(lib.rsv_poseidon2_permute(s.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), s.shape[0], device),
           "rsv_poseidon2_permute")
    return out


F_QADD, F_QSUB, F_QMUL, F_QINV, F_MMUL, F_MINV, F_CMUL, F_CINV, F_QMULI, F_QMULU, F_QPOW = range(11)  # RSV_F_*


def field_op(op: int, a, b=None, device: int = 0) -> np.ndarray:
    """Batched M31 / CM31 / QM31 arithmetic probe (rsv_field_op): a, b are (n, 4) QM31 words."""
    x = _u32(a).reshape(-1, 4)
    y = None if b is None else _u32(b).reshape(-1, 4)
    out = np.empty_like(x)
    _check(lib.rsv_field_op(op, x.ctypes.data_as(_u32p), None if y is None else y.ctypes.data_as(_u32p),
                            out.ctypes.data_as(_u32p), x.shape[0], device), "rsv_field_op")
    return out


def domain_points(log_size: int, q, device: int = 0) -> np.ndarray:
    """(n, 2) circle-domain points of the query positions q at log_size (rsv_domain_points)."""
    qq = _u32(q).reshape(-1)
    out = np.empty((qq.size, 2), np.uint32)
    _check(lib.rsv_domain_points(log_size, qq.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), qq.size, device),
           "rsv_domain_points")
    return out


def line_eval(coeffs, x, device: int = 0) -> np.ndarray:
    """LinePolyVar::eval_at_point of one polynomial (2^k QM31 coefficients) at the points x: (n, 4)."""
    c = _u32(coeffs).reshape(-1, 4)
    log_n = int(c.shape[0]).bit_length() - 1
    if c.shape[0] != 1 << log_n:
        raise ValueError("coefficient count must be a power of two")
    xx = _u32(x).reshape(-1)
    out = np.empty((xx.size, 4), np.uint32)
    _check(lib.rsv_line_eval(c.ctypes.data_as(_u32p), log_n, xx.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), xx.size,
                             device), "rsv_line_eval")
    return out


def oods_eval(samples, params, device: int = 0) -> np.ndarray:
    """a10 probe (rsv_oods_eval): samples (n, 142, 4), params (n, 26) = lp, lq, plonk_sum, poseidon_sum, z, alpha,
    random_coeff, oods.x -> (n, 8): constraint accumulator | expected value."""
    sm = _u32(samples).reshape(-1, 142, 4)
    pr = _u32(params).reshape(-1, 26)
    out = np.empty((sm.shape[0], 8), np.uint32)
    _check(lib.rsv_oods_eval(sm.ctypes.data_as(_u32p), pr.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p), sm.shape[0], device),
           "rsv_oods_eval")
    return out


def last_layer_check(coeffs, x, folded, device: int = 0) -> np.ndarray:
    """ok[i] = 1 iff LinePoly(coeffs)(x[i]) == folded[i] (rsv_last_layer_check; the RSV_R_FRI_LAST comparison)."""
    c = _u32(coeffs).reshape(-1, 4)
    log_n = int(c.shape[0]).bit_length() - 1
    xx = _u32(x).reshape(-1)
    f = _u32(folded).reshape(-1, 4)
    ok = np.zeros(xx.size, np.uint8)
    _check(lib.rsv_last_layer_check(c.ctypes.data_as(_u32p), log_n, xx.ctypes.data_as(_u32p), f.ctypes.data_as(_u32p),
                                    ok.ctypes.data_as(_u8p), xx.size, device), "rsv_last_layer_check")
    return ok


def half_permute(left, right, swap=None, device: int = 0):
    """Poseidon2HalfVar::permute: returns (rate, capacity), each (n, 8)."""
    l = _u32(left).reshape(-1, 8)
    r = _u32(right).reshape(-1, 8)
    n = l.shape[0]
    sw = None if swap is None else np.ascontiguousarray(swap, dtype=np.uint8)
    rate = np.empty((n, 8), np.uint32)
    cap = np.empty((n, 8), np.uint32)
    _check(lib.rsv_poseidon2_half_permute(l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p),
                                          None if sw is None else sw.ctypes.data_as(_u8p),
                                          rate.ctypes.data_as(_u32p), cap.ctypes.data_as(_u32p), n, device),
           "rsv_poseidon2_half_permute")
    return rate, cap


EMU_SWAP_ROWS, EMU_ROWS, EMU_STRIDE = 12, 413, 416  # RSV_EMU_SWAP_ROWS, RSV_EMU_ROWS, RSV_EMU_STRIDE


def poseidon2_emulated(left, right, swap=None, device: int = 0) -> np.ndarray:
    """Gate values of poseidon_permute_emulated for n permutations: (n, 416, 4) QM31 rows (rows 0..11 = the swap
    gates, zero for is_swap None; rows 409..412 = the output state; rows 413..415 zero padding).  swap: None or n bytes, 0 = None,
    1 = Some((false, _)), 2 = Some((true, _))."""
    l = _u32(left).reshape(-1, 8)
    r = _u32(right).reshape(-1, 8)
    n = l.shape[0]
    sw = None if swap is None else np.ascontiguousarray(swap, dtype=np.uint8)
    rows = np.empty((n, EMU_STRIDE, 4), np.uint32)
    _check(lib.rsv_poseidon2_emulated(l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p),
                                      None if sw is None else sw.ctypes.data_as(_u8p), rows.ctypes.data_as(_u32p), n,
                                      device), "rsv_poseidon2_emulated")
    return rows


def hash_node(children, cols, device: int = 0) -> np.ndarray:
    """Poseidon31MerkleHasher::hash_node for n nodes.  children = None or (left (n,8), right (n,8));
    cols = (n, n_cols) array (n_cols may be 0 when children are given)."""
    c = _u32(cols)
    if c.ndim == 1:
        c = c.reshape(1, -1)
    n, n_cols = c.shape
    out = np.empty((n, 8), np.uint32)
    if children is None:
        lp = rp = None
    else:
        l = _u32(children[0]).reshape(n, 8)
        r = _u32(children[1]).reshape(n, 8)
        lp, rp = l.ctypes.data_as(_u32p), r.ctypes.data_as(_u32p)
    _check(lib.rsv_merkle_hash_node(lp, rp, c.ctypes.data_as(_u32p) if n_cols else None, n_cols,
                                    out.ctypes.data_as(_u32p), n, device), "rsv_merkle_hash_node")
    return out


def merkle_path_root(query, siblings, cols, n_cols_at: Sequence[int], device: int = 0) -> np.ndarray:
    q = _u32(query).reshape(-1)
    n = q.shape[0]
    depth = len(n_cols_at) - 1
    sib = _u32(siblings).reshape(n, depth, 8) if depth else np.zeros((n, 0, 8), np.uint32)
    nca = _u32(n_cols_at)
    c = _u32(cols).reshape(n, int(nca.sum()))
    out = np.empty((n, 8), np.uint32)
    _check(lib.rsv_merkle_path_root(q.ctypes.data_as(_u32p), sib.ctypes.data_as(_u32p), c.ctypes.data_as(_u32p),
                                    nca.ctypes.data_as(_u32p), depth, out.ctypes.data_as(_u32p), n, device),
           "rsv_merkle_path_root")
    return out


def _parse_transcript(out: np.ndarray) -> dict:
    if out[0] == 1:
        return {"reason": "parse"}
    na, nq = int(out[1]), int(out[2])
    q = lambda o: tuple(int(x) for x in out[o:o + 4])
    return {
        "reason": REASONS[int(out[0])], "n_queries": nq, "log_size": int(out[3]),
        "z": q(4), "alpha": q(8), "random_coeff": q(12), "oods_t": q(16), "oods_x": q(20), "oods_y": q(24),
        "after_sampled_values_random_coeff": q(28), "pow_digest": [int(x) for x in out[32:40]],
        "fri_alphas": [q(40 + 4 * i) for i in range(na)],
        "raw_queries": [int(x) for x in out[40 + 4 * na:40 + 4 * na + nq]],
    }


def transcript(proof: bytes, device: int = 0) -> dict:
    b = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(1024, np.uint32)
    _check(lib.rsv_transcript(b.ctypes.data_as(_u8p), len(proof), out.ctypes.data_as(_u32p), out.size, device),
           "rsv_transcript")
    return _parse_transcript(out)


def pack(proofs: Sequence[bytes]):
    """Concatenate proofs into (blob uint8[total], offsets uint64[n+1])."""
    offsets = np.zeros(len(proofs) + 1, np.uint64)
    if proofs:
        offsets[1:] = np.cumsum([len(p) for p in proofs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(proofs), dtype=np.uint8) if proofs else np.zeros(0, np.uint8)
    return blob, offsets


def verify_batch(proofs: Sequence[bytes], cfg, inputs=STANDARD_INPUTS, device: int = 0):
    """Verify a batch of serialized proofs on the GPU under the configuration(s) `cfg` (required: one PcsConfig, or
    one per proof).  Returns (accept uint8[n], reason uint8[n])."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    pi = make_inputs(inputs)
    pc = prepare_cfg(cfg, n)
    _check(lib.rsv_verify_batch(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n,
                                pc.ref(), pi, len(list(inputs)),
                                accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), device), "rsv_verify_batch")
    return accept, reason


def trace_paths(proofs: Sequence[bytes], cfg, n_queries: int, max_log: int, inputs=STANDARD_INPUTS, device: int = 0):
    """per-query authentication paths of the four commitment trees, transcript query order.
    Returns (sib uint32[n,4,n_queries,max_log,8], pos uint32[n,4,n_queries], accept, reason)."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    sib = np.zeros((n, 4, n_queries, max_log, 8), np.uint32)
    pos = np.zeros((n, 4, n_queries), np.uint32)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    pi = make_inputs(inputs)
    _check(lib.rsv_trace_paths(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(cfg, n).ref(), pi, len(list(inputs)),
                               n_queries, max_log, sib.ctypes.data_as(_u32p), pos.ctypes.data_as(_u32p),
                               accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), device), "rsv_trace_paths")
    return sib, pos, accept, reason


def transcript_batch(proofs: Sequence[bytes], cfg, device: int = 0) -> np.ndarray:
    """FiatShamirHints of every proof of a batch (any mix of shapes): uint32[n, TRANSCRIPT_WORDS], layout in rsv.h."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    out = np.zeros((n, TRANSCRIPT_WORDS), np.uint32)
    _check(lib.rsv_transcript_batch(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(cfg, n).ref(), out.ctypes.data_as(_u32p),
                                    device), "rsv_transcript_batch")
    return out


def poseidon_flow_count(log_size_plonk: int, log_size_poseidon: int, cfg) -> int:
    """Records in the PoseidonFlow of one proof of this shape (rsv_poseidon_flow_count)."""
    c = PcsConfig(cfg.pow_bits, cfg.log_blowup_factor, cfg.log_last_layer_degree_bound, cfg.n_queries)
    out = ctypes.c_uint32(0)
    _check(lib.rsv_poseidon_flow_count(log_size_plonk, log_size_poseidon, ctypes.byref(c), ctypes.byref(out)), "rsv_poseidon_flow_count")
    return int(out.value)


def poseidon_flow(proofs: Sequence[bytes], cfg, flow_stride: int, inputs=STANDARD_INPUTS, device: int = 0):
    """SURVEY 8f.1 (second half): the PoseidonFlow of the circuit that verifies each proof, from the verifying pass
    (rsv_verify_hints with d_flow).  Returns (flow uint32[n, flow_stride, 32], swap uint8[n, flow_stride],
    count uint32[n], accept, reason); record = left8 | right8 | out_rate8 | out_cap8."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    flow = np.zeros((n, flow_stride, 32), np.uint32)
    swap = np.zeros((n, flow_stride), np.uint8)
    count = np.zeros(n, np.uint32)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    pi = make_inputs(inputs)
    ho = HintsOut(0, 0, 0, None, None, None, None, None, None, None, None, flow.ctypes.data, swap.ctypes.data, count.ctypes.data, flow_stride,
                  None, None)
    _check(lib.rsv_verify_hints(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(cfg, n).ref(), pi, len(list(inputs)),
                                ctypes.byref(ho), accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), device), "rsv_verify_hints")
    return flow, swap, count, accept, reason


def hints(proofs: Sequence[bytes], cfg, n_queries: int, max_log: int, n_inner: int, flow_stride: int, inputs=STANDARD_INPUTS, device: int = 0):
    """Everything the reference's hint structs hold for a uniform batch, from one verifying pass (rsv_verify_hints on host
    buffers): dict of trace_sib, trace_pos, trace_cols, fri_sib, fri_cols, flow, flow_swap, accept, reason (layouts: rsv.h)."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    out = {"trace_sib": np.zeros((n, 4, n_queries, max_log, 8), np.uint32), "trace_pos": np.zeros((n, 4, n_queries), np.uint32),
           "trace_cols": np.zeros((n, 4, n_queries, 64), np.uint32), "fri_sib": np.zeros((n, 1 + n_inner, n_queries, max_log, 8), np.uint32),
           "fri_cols": np.zeros((n, 1 + n_inner, n_queries, 3, 8), np.uint32), "flow": np.zeros((n, flow_stride, 32), np.uint32),
           "flow_swap": np.zeros((n, flow_stride), np.uint8), "accept": np.zeros(n, np.uint8), "reason": np.zeros(n, np.uint8)}
    pi = make_inputs(inputs)
    ho = HintsOut(n_queries, max_log, n_inner, None, out["trace_sib"].ctypes.data, out["trace_pos"].ctypes.data, out["trace_cols"].ctypes.data,
                  out["fri_sib"].ctypes.data, out["fri_cols"].ctypes.data, None, None, out["flow"].ctypes.data, out["flow_swap"].ctypes.data,
                  None, flow_stride, None, None)
    _check(lib.rsv_verify_hints(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(cfg, n).ref(), pi, len(list(inputs)),
                                ctypes.byref(ho), out["accept"].ctypes.data_as(_u8p), out["reason"].ctypes.data_as(_u8p), device), "rsv_verify_hints")
    return out


class WitnessProgram:
    """A witness program resident on the device (rsv_witness_program).  WitnessProgram.build(proof, cfg, ...) runs the
    library's mirror of the reference's circuit gadgets over a template proof (rsv_witness_program_build);
    WitnessProgram(program) loads arrays made earlier (witness_program.Program, e.g. from a file)."""

    def __init__(self, program=None, device: int = 0, _handle=None):
        self._h = None
        if _handle is not None:
            self._h = _handle
        else:
            sh = program.shape
            shape = WitnessShape(sh["lp"], sh["lq"], sh["pow_bits"], sh["blowup"], sh["log_last"], sh["nq"], sh["n_inner"], sh["flow_count"],
                                 sh["copies"])
            instr = np.ascontiguousarray(program.instr, dtype=np.uint32)
            levels = np.ascontiguousarray(program.level_offsets, dtype=np.uint32)
            h = ctypes.c_void_p()
            _check(lib.rsv_witness_program_create(instr.ctypes.data_as(_u32p), instr.shape[0], levels.ctypes.data_as(_u32p), len(levels) - 1,
                                                  program.n_vars, ctypes.byref(shape), device, ctypes.byref(h)), "rsv_witness_program_create")
            self._h = h
            self._flow_wires = program.flow_wires
        n_vars, n_levels = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self.shape = WitnessShape()
        _check(lib.rsv_witness_program_info(self._h, ctypes.byref(n_vars), ctypes.byref(n_levels), ctypes.byref(self.shape)),
               "rsv_witness_program_info")
        self.n_vars, self.n_levels = int(n_vars.value), int(n_levels.value)

    @classmethod
    def build(cls, proof: bytes, cfg, inputs=STANDARD_INPUTS, copies: int = 1, device: int = 0, set_walks=None) -> "WitnessProgram":
        """set_walks: per copy, which of the four orders the reference's two HashSet walks took (rsv.h); default 0 for all."""
        b = np.frombuffer(proof, dtype=np.uint8)
        walks = None if set_walks is None else np.ascontiguousarray(set_walks, dtype=np.uint8)
        if walks is not None and len(walks) != copies:
            raise ValueError("set_walks: one entry per copy")
        c = PcsConfig(cfg.pow_bits, cfg.log_blowup_factor, cfg.log_last_layer_degree_bound, cfg.n_queries)
        pi = make_inputs(inputs)
        h = ctypes.c_void_p()
        _check(lib.rsv_witness_program_build(b.ctypes.data_as(_u8p), len(proof), ctypes.byref(c), pi, len(list(inputs)), copies,
                                             walks.ctypes.data_as(_u8p) if walks is not None else None, device, ctypes.byref(h)),
               "rsv_witness_program_build")
        return cls(_handle=h)

    def export(self):
        """-> witness_program.Program (instr, level_offsets, shape, flow_wires): what save / save_raw write and __init__ takes back."""
        instr = np.zeros((self.n_vars, 8), np.uint32)
        levels = np.zeros(self.n_levels + 1, np.uint32)
        s = self.shape
        wires = getattr(self, "_flow_wires", None)
        if wires is None:
            wires = np.zeros((s.copies * s.flow_count, 5), np.uint32)
            _check(lib.rsv_witness_program_export(self._h, instr.ctypes.data_as(_u32p), levels.ctypes.data_as(_u32p), wires.ctypes.data_as(_u32p)),
                   "rsv_witness_program_export")
        else:
            _check(lib.rsv_witness_program_export(self._h, instr.ctypes.data_as(_u32p), levels.ctypes.data_as(_u32p), None),
                   "rsv_witness_program_export")
        shape = dict(zip(witness_program.SHAPE_KEYS, (s.log_size_plonk, s.log_size_poseidon, s.pow_bits, s.log_blowup, s.log_last, s.n_queries,
                                              s.n_inner, s.flow_count, s.copies)))
        return witness_program.Program(instr, levels, self.n_vars, shape, wires)

    def gates(self, variables=None):
        """The circuit's Plonk rows (rsv_witness_program_gates): uint32[n_rows, 6] = a_wire, b_wire, c_wire, op,
        poseidon_wire, enforce_c_m31 — the template's; with `variables` (uint32[n_vars, 4] of another proof of the shape) the
        rows whose op follows the witness are set for that proof.  Also returns witness_ops uint32[n, 3]."""
        n_rows, n_ops = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _check(lib.rsv_witness_program_gates(self._h, ctypes.byref(n_rows), ctypes.byref(n_ops), None, None), "rsv_witness_program_gates")
        rows = np.zeros((n_rows.value, 6), np.uint32)
        ops = np.zeros((n_ops.value, 3), np.uint32)
        _check(lib.rsv_witness_program_gates(self._h, None, None, rows.ctypes.data_as(_u32p), ops.ctypes.data_as(_u32p)), "rsv_witness_program_gates")
        if variables is not None:
            rows[ops[:, 0], 3] = np.where(variables[ops[:, 1], 0] != 0, ops[:, 2], 0)
        return rows, ops

    def close(self):
        if getattr(self, "_h", None):
            lib.rsv_witness_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cfg(self) -> PcsConfig:
        s = self.shape
        return PcsConfig(s.pow_bits, s.log_blowup, s.log_last, s.n_queries)

    def scratch_bytes(self, n: int) -> int:
        out = ctypes.c_size_t(0)
        _check(lib.rsv_witness_scratch_bytes(self._h, n, ctypes.byref(out)), "rsv_witness_scratch_bytes")
        return int(out.value)


def witness(proofs: Sequence[bytes], program: WitnessProgram, inputs=STANDARD_INPUTS, device: int = 0, with_flow: bool = False):
    """`variables` of the recursion circuit for every proof of a batch (rsv_witness_eval): uint32[n, n_vars, 4], accept, reason
    [, flow uint32[n, flow_count, 32], flow_swap uint8[n, flow_count] with with_flow]."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    variables = np.zeros((n, program.n_vars, 4), np.uint32)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    flow = np.zeros((n, program.shape.flow_count, 32), np.uint32) if with_flow else None
    swap = np.zeros((n, program.shape.flow_count), np.uint8) if with_flow else None
    pi = make_inputs(inputs)
    _check(lib.rsv_witness_eval(program._h, blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(program.cfg(), n).ref(), pi,
                                len(list(inputs)), variables.ctypes.data_as(_u32p), flow.ctypes.data_as(_u32p) if with_flow else None,
                                swap.ctypes.data_as(_u8p) if with_flow else None, accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), device),
           "rsv_witness_eval")
    return (variables, accept, reason, flow, swap) if with_flow else (variables, accept, reason)


def fri_paths(proofs: Sequence[bytes], cfg, n_queries: int, max_log: int, n_inner: int, inputs=STANDARD_INPUTS, device: int = 0):
    """per-query pair paths of the FRI trees.  Returns (sib uint32[n,1+n_inner,nq,max_log,8],
    cols uint32[n,1+n_inner,nq,3,8], accept, reason)."""
    blob, offsets = pack(proofs)
    n = len(proofs)
    sib = np.zeros((n, 1 + n_inner, n_queries, max_log, 8), np.uint32)
    cols = np.zeros((n, 1 + n_inner, n_queries, 3, 8), np.uint32)
    accept = np.zeros(n, np.uint8)
    reason = np.zeros(n, np.uint8)
    pi = make_inputs(inputs)
    _check(lib.rsv_fri_paths(blob.ctypes.data_as(_u8p), offsets.ctypes.data_as(_u64p), n, prepare_cfg(cfg, n).ref(), pi, len(list(inputs)), n_queries,
                             max_log, n_inner, sib.ctypes.data_as(_u32p), cols.ctypes.data_as(_u32p),
                             accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), device), "rsv_fri_paths")
    return sib, cols, accept, reason


class HostArena:
    """rsv_host_alloc: pinned host memory the caller reads its proofs into, back to back; `buf` is a numpy uint8 view of it.
    pack(proofs) copies a list of proofs in and returns the HostBatch over the copies (what a caller that deserialises
    straight into the arena would hold)."""

    def __init__(self, nbytes: int):
        h = ctypes.c_void_p()
        _check(lib.rsv_host_alloc(nbytes, ctypes.byref(h)), "rsv_host_alloc")
        self._h = h
        self.nbytes = nbytes
        self.buf = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(h.value))

    def pack(self, proofs) -> "HostBatch":
        views, at = [], 0
        for p in proofs:
            a = np.frombuffer(p, dtype=np.uint8) if not isinstance(p, np.ndarray) else p
            if at + a.size > self.nbytes:
                raise ValueError("arena too small")
            self.buf[at:at + a.size] = a
            views.append(self.buf[at:at + a.size])
            at += a.size
        return HostBatch(views)

    def close(self):
        if self._h:
            self.buf = None
            lib.rsv_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatch:
    """The argument of rsv_verify_batch_host as a Rust caller holds it (a Vec of serialized proofs = pointers + lengths): built
    once from a list of bytes / numpy uint8 buffers, which it keeps alive."""

    def __init__(self, proofs):
        self.keep = [np.frombuffer(p, dtype=np.uint8) if not isinstance(p, np.ndarray) else p for p in proofs]
        self.n = len(self.keep)
        self.ptrs = (ctypes.c_void_p * max(self.n, 1))(*[k.ctypes.data for k in self.keep])
        self.lens = np.array([k.size for k in self.keep], dtype=np.uint64)
        self.bytes = int(self.lens.sum())


class Context:
    """One HIP stream + reusable HBM workspace on one device (rsv_ctx).  Operates on torch tensors that
    already live on that device; nothing is copied through the host.

    Stream ordering: the context enqueues on its own non-blocking streams.  Every method that reads or writes
    caller tensors first makes the context wait for torch's CURRENT stream (rsv_ctx_wait_stream), so tensors
    produced by torch kernels / copies immediately before the call are complete when the verifier reads them, and
    output buffers freshly allocated or zeroed by torch are not overwritten early.  Results are ordered for torch
    by `synchronize()` (host blocks) or `release_to_torch()` (torch's current stream waits, host does not)."""

    def __init__(self, device: int = 0):
        h = ctypes.c_void_p()
        _check(lib.rsv_ctx_create(device, ctypes.byref(h)), "rsv_ctx_create")
        self._h = h
        self.device = device
        self._live = []
        self._dependents = []  # weak references to the Exchange objects created on this context

    def close(self):
        if self._h:
            for ref in self._dependents:  # an exchange enqueues on this context's stream: it goes first (rsv.h)
                x = ref()
                if x is not None:
                    x.close()
            self._dependents = []
            lib.rsv_ctx_destroy(self._h)  # waits for the context's streams
            self._h = None
            self._live = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib.rsv_ctx_synchronize(self._h), "rsv_ctx_synchronize")
        self._live = []

    def set_option(self, name: str, value) -> None:
        """A tuning / diagnostic knob of this context (rsv_ctx_set_option; names in OPTIONS)."""
        _set_option(self._h, name, value)

    def _keep(self, pc, cfg):
        """A per-proof configuration index uploaded INSIDE a call (cfg was a list, not a PreparedCfg the caller holds)
        lives in a torch tensor that the call's first kernel reads later, on this context's stream.  Were it to die with
        the call, torch's caching allocator could hand its memory to the next allocation on torch's stream, which is not
        ordered behind the context's.  So the context holds it: until synchronize() / close(), or — when many calls are
        queued without one — until torch's current stream has been made to wait for the context (release_to_torch), after
        which a reuse of the block is ordered behind every kernel that read it.  (A PreparedCfg the caller made is the
        caller's to keep alive until the work is ordered, like the blob.)"""
        if pc is cfg:
            return
        self._live.append(pc)
        if len(self._live) > 16:
            self.release_to_torch()
            self._live = self._live[-1:]  # the newest belongs to the call that is about to be enqueued

    @property
    def stream(self) -> int:
        return lib.rsv_ctx_stream(self._h) or 0

    def _torch_stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def acquire_from_torch(self):
        """Work enqueued so far on torch's current stream happens before what this context enqueues next."""
        _check(lib.rsv_ctx_wait_stream(self._h, self._torch_stream()), "rsv_ctx_wait_stream")

    def release_to_torch(self):
        """torch's current stream waits for everything this context has enqueued so far (no host block)."""
        _check(lib.rsv_stream_wait_ctx(self._h, self._torch_stream()), "rsv_stream_wait_ctx")

    def prepare_cfg(self, cfg, n: int) -> PreparedCfg:
        """Stage a configuration set for repeated use (the per-proof index of a mixed batch goes to HBM once)."""
        def to_dev(a):
            import torch
            return torch.from_numpy(a).to(torch.device("cuda", self.device))
        return prepare_cfg(cfg, n, to_dev)

    def poseidon2_permute(self, d_in, d_out, d_bad=None):
        n = d_in.numel() // 16
        self.acquire_from_torch()
        _check(lib.rsv_poseidon2_permute_dev(self._h, d_in.data_ptr(), d_out.data_ptr(), n,
                                             d_bad.data_ptr() if d_bad is not None else None), "rsv_poseidon2_permute_dev")

    def poseidon2_emulated(self, d_left, d_right, d_swap, d_rows, d_bad=None):
        """d_left / d_right: (n, 8) u32; d_swap: None or (n,) u8; d_rows: (n, 416, 4) u32 (rsv_poseidon2_emulated_dev)."""
        n = d_left.numel() // 8
        self.acquire_from_torch()
        _check(lib.rsv_poseidon2_emulated_dev(self._h, d_left.data_ptr(), d_right.data_ptr(),
                                              d_swap.data_ptr() if d_swap is not None else None, d_rows.data_ptr(), n,
                                              d_bad.data_ptr() if d_bad is not None else None), "rsv_poseidon2_emulated_dev")

    def verify_batch(self, d_blob, d_offsets, n: int, d_accept, d_reason=None, cfg=None, inputs=STANDARD_INPUTS):
        pi = make_inputs(inputs)
        pc = self.prepare_cfg(cfg, n)
        self.acquire_from_torch()
        self._keep(pc, cfg)
        _check(lib.rsv_verify_batch_dev(self._h, d_blob.data_ptr(), d_offsets.data_ptr(), n, pc.ref(), pi, len(list(inputs)),
                                        d_accept.data_ptr(), d_reason.data_ptr() if d_reason is not None else None),
               "rsv_verify_batch_dev")

    def trace_paths(self, d_blob, d_offsets, n: int, n_queries: int, max_log: int, d_sib, d_pos, d_accept,
                    d_reason=None, cfg=None, inputs=STANDARD_INPUTS):
        pi = make_inputs(inputs)
        pc = self.prepare_cfg(cfg, n)
        self.acquire_from_torch()
        self._keep(pc, cfg)
        _check(lib.rsv_trace_paths_dev(self._h, d_blob.data_ptr(), d_offsets.data_ptr(), n, pc.ref(), pi, len(list(inputs)),
                                       n_queries, max_log, d_sib.data_ptr(), d_pos.data_ptr(), d_accept.data_ptr(),
                                       d_reason.data_ptr() if d_reason is not None else None), "rsv_trace_paths_dev")

    def verify_batch_host(self, proofs, cfg, inputs=STANDARD_INPUTS):
        """Proofs in host memory, one buffer each (bytes / numpy uint8 arrays, or a HostBatch that holds the pointer table of
        such a list already): gather, upload and verify overlap (rsv_verify_batch_host).  Returns (accept, reason) numpy arrays."""
        hb = proofs if isinstance(proofs, HostBatch) else HostBatch(proofs)
        n = hb.n
        accept = np.zeros(n, np.uint8)
        reason = np.zeros(n, np.uint8)
        pi = make_inputs(inputs)
        pc = prepare_cfg(cfg, n)  # host residency: the library uploads the per-proof index chunk by chunk
        _check(lib.rsv_verify_batch_host(self._h, hb.ptrs, hb.lens.ctypes.data_as(_u64p), n, pc.ref(), pi, len(list(inputs)),
                                         accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p)), "rsv_verify_batch_host")
        return accept, reason

    def verify_hints(self, d_blob, d_offsets, n: int, d_accept, d_reason=None, cfg=None, inputs=STANDARD_INPUTS, shape=(0, 0, 0),
                     d_transcript=None, d_trace_sib=None, d_trace_pos=None, d_trace_cols=None, d_fri_sib=None,
                     d_fri_cols=None, d_fri_folded=None, d_query_values=None, d_flow=None, d_flow_swap=None, d_flow_count=None,
                     d_accept_bitmap=None, d_accept_count=None):
        """One verifying pass that also fills whichever hint outputs are given (rsv_verify_hints_dev).
        shape = (n_queries, max_log, n_inner), needed for the path outputs."""
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        ho = HintsOut(int(shape[0]), int(shape[1]), int(shape[2]), ptr(d_transcript), ptr(d_trace_sib), ptr(d_trace_pos),
                      ptr(d_trace_cols), ptr(d_fri_sib), ptr(d_fri_cols), ptr(d_fri_folded), ptr(d_query_values),
                      ptr(d_flow), ptr(d_flow_swap), ptr(d_flow_count), int(d_flow.shape[1]) if d_flow is not None else 0,
                      ptr(d_accept_bitmap), ptr(d_accept_count))
        pi = make_inputs(inputs)
        pc = self.prepare_cfg(cfg, n)
        self.acquire_from_torch()
        self._keep(pc, cfg)
        _check(lib.rsv_verify_hints_dev(self._h, d_blob.data_ptr(), d_offsets.data_ptr(), n, pc.ref(), pi, len(list(inputs)),
                                        ctypes.byref(ho), d_accept.data_ptr(), ptr(d_reason)), "rsv_verify_hints_dev")

    def witness(self, program: WitnessProgram, d_blob, d_offsets, n: int, d_variables, d_accept, d_reason=None, inputs=STANDARD_INPUTS,
                d_flow=None, d_flow_swap=None):
        """rsv_witness_eval_dev: d_variables uint32[n, n_vars, 4] in HBM (optionally the PoseidonFlow: d_flow uint32[n, flow_count,
        32], d_flow_swap uint8[n, flow_count]); enqueued on the context's streams."""
        pi = make_inputs(inputs)
        pc = self.prepare_cfg(program.cfg(), n)
        self.acquire_from_torch()
        self._keep(pc, None)
        _check(lib.rsv_witness_eval_dev(self._h, program._h, d_blob.data_ptr(), d_offsets.data_ptr(), n, pc.ref(), pi, len(list(inputs)),
                                        d_variables.data_ptr(), d_flow.data_ptr() if d_flow is not None else None,
                                        d_flow_swap.data_ptr() if d_flow_swap is not None else None, d_accept.data_ptr(),
                                        d_reason.data_ptr() if d_reason is not None else None),
               "rsv_witness_eval_dev")

    def accept_bitmap(self, d_accept, n: int, d_bitmap, d_count=None):
        self.acquire_from_torch()
        _check(lib.rsv_accept_bitmap_dev(self._h, d_accept.data_ptr(), n, d_bitmap.data_ptr(),
                                         d_count.data_ptr() if d_count is not None else None), "rsv_accept_bitmap_dev")

    def last_stage_times(self) -> dict:
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        k = lib.rsv_last_stage_times(self._h, names, ms, 16)
        if k < 0:
            raise RsvError(k, "rsv_last_stage_times")
        return {names[i].decode(): float(ms[i]) for i in range(k)}


def shard_range(n_total: int, rank: int, world: int):
    """rsv_shard_range: the contiguous [lo, hi) of rank `rank` (the rule sharding.shard_range restates in Python)."""
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    lib.rsv_shard_range(n_total, rank, world, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def shard_plan(lens, world: int):
    """rsv_shard_plan: contiguous cuts balanced by bytes.  lens: the job's proof lengths.  Returns (lo, hi), `world` entries each."""
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    lo, hi = (ctypes.c_size_t * world)(), (ctypes.c_size_t * world)()
    _check(lib.rsv_shard_plan(ln.ctypes.data_as(_u64p), len(ln), world, lo, hi), "rsv_shard_plan")
    return list(lo), list(hi)


def cfg_check(cfg: "PcsConfig") -> bool:
    """rsv_cfg_check: is the configuration inside the library's shape limits (include/rsv.h: RSV_MAX_*)?"""
    return lib.rsv_cfg_check(ctypes.byref(cfg)) == 0


class MultiContext:
    """rsv_multi: ONE process drives several contexts (one per entry of `devices`; a device may repeat), one host thread
    per context per call; the job's verdicts, bitmap and count are assembled on the host (no collective)."""

    def __init__(self, devices: Sequence[int]):
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        _check(lib.rsv_multi_create(arr, len(devices), ctypes.byref(h)), "rsv_multi_create")
        self._h = h
        self.devices = list(devices)

    def close(self):
        if self._h:
            lib.rsv_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib.rsv_multi_size(self._h))

    def set_option(self, name: str, value) -> None:
        for r in range(len(self)):
            _set_option(ctypes.c_void_p(lib.rsv_multi_ctx(self._h, r)), name, value)

    def verify_batch_host(self, proofs, cfg, inputs=STANDARD_INPUTS):
        """The whole job in host memory (bytes / numpy buffers or a HostBatch).  Returns (accept, reason, bitmap, count)."""
        hb = proofs if isinstance(proofs, HostBatch) else HostBatch(proofs)
        n = hb.n
        accept, reason = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        bitmap = np.zeros(max(1, (n + 31) // 32), np.uint32)
        count = ctypes.c_uint64()
        pi = make_inputs(inputs)
        pc = prepare_cfg(cfg, n)
        _check(lib.rsv_multi_verify_batch_host(self._h, hb.ptrs, hb.lens.ctypes.data_as(_u64p), n, pc.ref(), pi, len(list(inputs)),
                                               accept.ctypes.data_as(_u8p), reason.ctypes.data_as(_u8p), bitmap.ctypes.data_as(_u32p),
                                               ctypes.byref(count)), "rsv_multi_verify_batch_host")
        return accept, reason, bitmap[: (n + 31) // 32], int(count.value)

    def verify_batch_dev(self, shards, cfg_table: Sequence[PcsConfig], inputs=STANDARD_INPUTS):
        """shards: one dict per context with torch tensors on that context's device: d_blob, d_offsets, n, and optionally
        d_cfg_of / d_accept / d_reason.  The caller has synchronised whatever produced them (the call waits for nothing
        of torch's).  Returns (bitmap uint32[ceil(sum n / 32)], count)."""
        if len(shards) != len(self):
            raise ValueError("one shard per context")
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        arr = (Shard * len(shards))(*[Shard(ptr(s["d_blob"]), ptr(s["d_offsets"]), int(s["n"]), ptr(s.get("d_cfg_of")),
                                            ptr(s.get("d_accept")), ptr(s.get("d_reason"))) for s in shards])
        n_total = sum(int(s["n"]) for s in shards)
        bitmap = np.zeros(max(1, (n_total + 31) // 32), np.uint32)
        count = ctypes.c_uint64()
        pi = make_inputs(inputs)
        pc = PreparedCfg(list(cfg_table))
        _check(lib.rsv_multi_verify_batch_dev(self._h, arr, len(shards), pc.ref(), pi, len(list(inputs)), bitmap.ctypes.data_as(_u32p),
                                              ctypes.byref(count)), "rsv_multi_verify_batch_dev")
        return bitmap[: (n_total + 31) // 32], int(count.value)


def exchange_available() -> bool:
    return bool(lib.rsv_exchange_available())


def exchange_unique_id() -> bytes:
    buf = (ctypes.c_uint8 * EXCHANGE_ID_BYTES)()
    _check(lib.rsv_exchange_unique_id(buf), "rsv_exchange_unique_id")
    return bytes(buf)


def exchange_assemble(n_total: int, world: int, gathered: np.ndarray, plan=None):
    """rsv_exchange_assemble[_plan]: gathered [world][slice_words] (host) -> (accept bytes, contiguous bitmap) of the job.
    plan = (lo, hi) of every rank for a job cut by rsv_shard_plan (or by the caller); None: rsv_shard_range's cuts."""
    g = np.ascontiguousarray(gathered, dtype=np.uint32)
    accept = np.zeros(max(n_total, 1), np.uint8)
    bitmap = np.zeros(max(1, (n_total + 31) // 32), np.uint32)
    if plan is None:
        _check(lib.rsv_exchange_assemble(n_total, world, g.ctypes.data_as(_u32p), accept.ctypes.data_as(_u8p), bitmap.ctypes.data_as(_u32p)),
               "rsv_exchange_assemble")
    else:
        lo, hi = (ctypes.c_size_t * world)(*plan[0]), (ctypes.c_size_t * world)(*plan[1])
        _check(lib.rsv_exchange_assemble_plan(world, lo, hi, g.ctypes.data_as(_u32p), accept.ctypes.data_as(_u8p), bitmap.ctypes.data_as(_u32p)),
               "rsv_exchange_assemble_plan")
    return accept[:n_total], bitmap[: (n_total + 31) // 32]


class Exchange:
    """rsv_exchange: the one-process-per-GPU collective step over RCCL, driven through the C-ABI (no torch.distributed on
    the data path).  `uid` = exchange_unique_id() of rank 0, distributed by the caller.  Collective constructor."""

    def __init__(self, ctx: "Context", uid: bytes, rank: int, world: int, n_total: int, plan=None):
        """plan = (lo, hi) of every rank (rsv_shard_plan's cuts, or the caller's): rsv_exchange_create_plan; None: the job is
        cut by rsv_shard_range.  Close the exchange before its context (rsv.h)."""
        buf = (ctypes.c_uint8 * EXCHANGE_ID_BYTES).from_buffer_copy(uid)
        h = ctypes.c_void_p()
        if plan is None:
            _check(lib.rsv_exchange_create(ctx._h, buf, rank, world, n_total, ctypes.byref(h)), "rsv_exchange_create")
        else:
            lo_a, hi_a = (ctypes.c_size_t * world)(*plan[0]), (ctypes.c_size_t * world)(*plan[1])
            _check(lib.rsv_exchange_create_plan(ctx._h, buf, rank, world, lo_a, hi_a, ctypes.byref(h)), "rsv_exchange_create_plan")
        self._h, self.ctx, self.rank, self.world, self.n_total = h, ctx, rank, world, n_total
        ctx._dependents.append(weakref.ref(self))
        lo, hi, sw = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _check(lib.rsv_exchange_layout(h, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(sw)), "rsv_exchange_layout")
        self.lo, self.hi, self.slice_words = lo.value, hi.value, sw.value

    def run(self, d_local, d_gathered, d_count=None):
        _check(lib.rsv_exchange_run(self._h, d_local.data_ptr(), d_gathered.data_ptr(), d_count.data_ptr() if d_count is not None else None),
               "rsv_exchange_run")

    def close(self):
        if self._h:
            lib.rsv_exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import witness_program  # noqa: E402,F401  (the witness program container)
