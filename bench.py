#!/usr/bin/env python3
"""bench.py — recursive proofs verified / s on MI355X (BASELINE.json metric).

A "step" is one pass of the verify pipeline over one synthetic batch that is already resident in HBM:
BASELINE configs[2], 65 536 proofs built round-robin from the reference's four standard-config fixtures
(recursive_proof_16_15, level3-1, level6-1, level7-1) with the seeded tamper rule of SURVEY §8d
(proof i with i % 17 == 5 gets one flipped bit), verified under the reference's `standard_config`
(examples/multi-proofs/src/main.rs:173-176).

Multi-GPU (`--gpus N`): one process per GPU.  Rank r verifies the contiguous shard r of the job and the ranks
exchange their accept bitmaps with one all-gather + one all-reduce of the count per step over RCCL
(recursive-stwo_amd/sharding.py: ShardedVerifier.step — the same function the tests cover).
  * default: every rank gets `--proofs` proofs (65 536)                      -> "scaling": "weak"
  * `--total-proofs T` (BASELINE configs[3]: 1 048 576): the job is T proofs, split over the ranks -> "strong"
Rank 0 prints ONE JSON line.

Launch: `python bench.py [--gpus N]` — for N > 1 this process only starts N rank processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, as children, before
anything touches a GPU) and waits for them; the explicit torchrun form the driver uses works as well.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# dmabuf IPC: RCCL across rank processes needs it on this driver.  Set here, before anything can initialise HIP, so
# that ranks started by an external torchrun (the driver's form) get it as well as the ones launch_ranks() starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURES = ["recursive_proof_16_15.bin", "level3-1.bin", "level6-1.bin", "level7-1.bin"]
# other workloads (parity-test configurations of BASELINE.json, selectable with --workload; not the bench line)
WORKLOADS = {
    "standard": FIXTURES,                                   # BASELINE configs[2]
    "copies": ["recursive_proof_16_15.bin"],                # BASELINE configs[1]: copies of one proof
    "chain": ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin",
              "level7-1.bin", "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin",
              "level13-1.bin"],                             # BASELINE configs[4]: recursion chain, mixed shapes
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def splitmix64(seed, i):
    mask = (1 << 64) - 1
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & mask
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    return z ^ (z >> 31)


def read_fixture(name):
    with open(os.path.join(ROOT, "tests", "golden", "proofs", name), "rb") as f:
        return f.read()


def fixture_configs(rsv, fixtures):
    """The PcsConfig literal the reference verifies each fixture under (tests/golden/manifest.json cites the source
    line of every one: examples/multi-proofs/src/main.rs:173-295)."""
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        man = {e["file"]: e for e in json.load(f)["proofs"]}
    return [rsv.PcsConfig(man[f]["pow_bits"], man[f]["log_blowup_factor"], man[f]["log_last_layer_degree_bound"],
                          man[f]["n_queries"]) for f in fixtures]


def kernel_sources_sha():
    """Identity of the kernels a profile was taken on (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "recursive-stwo_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip", ".inc")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def job_fixture_index(i, n_fixtures, order="round_robin", n_total=None):
    """Which fixture proof i of the job is a copy of.  round_robin: i mod F (the bench mixes).  level: the job in the order
    the reference's driver produces it (examples/multi-proofs/src/main.rs:198-295): all copies of level 1, then level 2, ...
    — equal counts per fixture, the last one takes the remainder."""
    i = np.asarray(i, dtype=np.int64)
    if order == "round_robin":
        return i % n_fixtures
    per = max(1, int(n_total) // n_fixtures)
    return np.minimum(i // per, n_fixtures - 1)


def job_lengths(n_total, fixtures, order):
    """Byte length of every proof of the job (what rsv_shard_plan balances): every rank computes the same plan from it."""
    lens = np.array([len(read_fixture(f)) for f in fixtures], dtype=np.uint64)
    return lens[job_fixture_index(np.arange(n_total), len(fixtures), order, n_total)]


def build_batch_on_device(torch, dev, n_proofs, first_index, fixtures=FIXTURES, order="round_robin", n_total=None):
    """Returns (d_blob uint8, d_offsets int64[n+1], lengths, tamper indices, fixture index per proof) for global proof
    indices [first_index, first_index + n_proofs)."""
    proofs = [read_fixture(f) for f in fixtures]
    lens = np.array([len(p) for p in proofs], dtype=np.int64)
    idx = job_fixture_index(np.arange(n_proofs, dtype=np.int64) + first_index, len(proofs), order, n_total)
    plen = lens[idx]
    offsets = np.zeros(n_proofs + 1, dtype=np.int64)
    np.cumsum(plen, out=offsets[1:])
    total = int(offsets[-1])
    if order == "round_robin":
        # one period (aligned to the round-robin phase) tiled across the batch, built in HBM
        phase = int(first_index % len(proofs))
        seq = [(phase + k) % len(proofs) for k in range(len(proofs))]
        period = np.frombuffer(b"".join(proofs[k] for k in seq), dtype=np.uint8)
        d_period = torch.from_numpy(period.copy()).to(dev)
        reps = (n_proofs + len(proofs) - 1) // len(proofs)
        d_blob = d_period.repeat(reps)[:total].contiguous()
    else:
        # runs of one fixture each, built in HBM
        cuts = np.flatnonzero(np.diff(idx)) + 1
        starts = np.concatenate([[0], cuts, [n_proofs]])
        parts = []
        for a, b in zip(starts[:-1], starts[1:]):
            if b > a:
                parts.append(torch.from_numpy(np.frombuffer(proofs[int(idx[a])], dtype=np.uint8).copy()).to(dev).repeat(int(b - a)))
        d_blob = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
    # seeded tampering
    tam = [i for i in range(n_proofs) if (first_index + i) % 17 == 5]
    pos = np.array([offsets[i] + 60 + splitmix64(0xC0FFEE, first_index + i) % (int(plen[i]) - 68) for i in tam],
                   dtype=np.int64)
    if len(pos):
        d_pos = torch.from_numpy(pos).to(dev)
        d_blob[d_pos] = d_blob[d_pos] ^ 1
    d_offsets = torch.from_numpy(offsets).to(dev)
    return d_blob, d_offsets, plen, np.array(tam, dtype=np.int64), idx


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """What the box lets this process use: logical CPUs, scheduler affinity, cgroup CPU quota.  A one-GPU slice of a
    256-core host is capped well below os.cpu_count(): on the round-4 box cpu.max = "1600000 100000" = 16 CPUs, and 256
    oracle threads verify FEWER proofs per second than 16 (3 231 against 4 107: the quota throttles them)."""
    info = {"os_cpu_count": os.cpu_count()}
    usable = os.cpu_count() or 1
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
        usable = min(usable, info["sched_affinity"])
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            info["cgroup_cpu_max"] = f.read().strip()
        quota, period = info["cgroup_cpu_max"].split()
        if quota != "max":
            usable = min(usable, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            info["cgroup_cfs_quota_us"], info["cgroup_cfs_period_us"] = q, per
            if q > 0:
                usable = min(usable, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    info["usable_cpus"] = usable
    return info


def cpu_baseline(blob_host, offsets, n_sample, cfg_rows, cfg_of, fixtures=None):
    """The C oracle (a port of the reference algorithm, not the Rust binary) on the host cores ("all host
    cores").  Three legs, about 160 proofs per thread each (1-1.5 s):
      value              every CPU this process may USE: min(logical CPUs, scheduler affinity, cgroup quota) threads
      all_logical_cpus   os.cpu_count() threads, when that is more than the usable share (reported, so that nobody has
                         to wonder what 256 threads would have given: less, the quota throttles them)
      one_thread         one thread."""
    import ctypes
    from tests import oracle_binding as ob
    share = cpu_share()
    threads = share["usable_cpus"]
    n_sample = min(n_sample, len(offsets) - 1)
    pi = ob.make_inputs(ob.STANDARD_INPUTS)
    arr = (ob.PcsConfig * len(cfg_rows))(*[ob.PcsConfig(*r) for r in cfg_rows])
    of = np.ascontiguousarray(cfg_of[:n_sample], dtype=np.uint8)

    def run(n_run, n_threads):
        n_threads = max(1, min(n_threads, n_run))
        bounds = np.linspace(0, n_run, n_threads + 1).astype(int)

        def work(t):
            lo, hi = bounds[t], bounds[t + 1]
            if hi <= lo:
                return 0
            offs = np.ascontiguousarray(offsets[lo:hi + 1], dtype=np.uint64)
            acc = np.zeros(hi - lo, np.uint8)
            cs = ob.CfgSet(ctypes.cast(arr, ctypes.POINTER(ob.PcsConfig)), len(cfg_rows), of[lo:hi].ctypes.data)
            rc = ob.lib.rsvo_verify_batch(blob_host.ctypes.data_as(ob._u8p), offs.ctypes.data_as(ob._u64p), hi - lo, ctypes.byref(cs),
                                          pi, 3, acc.ctypes.data_as(ob._u8p), None)
            assert rc == 0
            return int(acc.sum())

        t0 = time.perf_counter()
        with ThreadPoolExecutor(n_threads) as ex:
            accepted = sum(ex.map(work, range(n_threads)))
        return time.perf_counter() - t0, accepted, n_threads

    n_use = max(1, min(n_sample, 640 * threads))
    dt, accepted, used = run(n_use, threads)
    all_logical = None
    logical = os.cpu_count() or 1
    if logical > threads and n_sample >= 2 * threads:
        n_all = max(1, min(n_sample, 40 * logical))
        dt_all, _, used_all = run(n_all, logical)
        all_logical = {"value": n_all / dt_all, "unit": "proofs/s", "cores": used_all, "proofs": n_all,
                       "note": "one thread per logical CPU the host reports; the cgroup quota above throttles them"}
    n_one = max(1, min(n_sample, 160))
    dt_one, _, _ = run(n_one, 1)
    # Poseidon2 permutations the oracle's batched walk spends on the genuine fixtures of this workload (untimed): the
    # unit of useful work of this path, used for the in-situ permutation rate reported next to the microbenchmark
    perms = [ob.perm_count(read_fixture(f)) for f in fixtures] if fixtures else []
    return {"value": n_use / dt, "unit": "proofs/s", "cores": used, "kind": "port",
            "sample": f"first {n_use} proofs of the rank-0 batch ({accepted} accepted), C oracle, {used} threads = every CPU this "
                      f"process may use (cpu_share: {logical} logical CPUs, usable {threads}); one_thread: the first {n_one} on 1 thread",
            "all_logical_cpus": all_logical,
            "one_thread": {"value": n_one / dt_one, "unit": "proofs/s", "cores": 1},
            "cpu_model": cpu_model(), "host_cores": logical, "cpu_share": share,
            "note": "oracle/rsv_oracle.c: a plain scalar C restatement of the reference algorithm (not the Rust binary, which "
                    "cannot be built here; Mersenne-fold modular multiply, no SIMD); a reported baseline, not a tuned CPU implementation",
            "perms_per_proof": (sum(perms) / len(perms)) if perms else None}


# Instruction-cost ceiling of the permutation on one MI355X: the dynamic VALU mix of one wave-level call of
# rsv::poseidon2() (64 permutations) priced with the per-instruction issue costs measured by tools/valu_lab.hip at
# 4 waves/SIMD.  Derivation, class table and the PMC cross-check: tools/perm_ceiling.py (same numbers) and DESIGN §4.
PERM_MIX = [("fast", 2456, 2.50), ("v_min_u32", 442, 4.27), ("v_mad_u64_u32", 526, 4.54), ("v_lshl_add_u64", 410, 4.48),
            ("v_mad_u64_u32+addend", 564, 5.10)]
SIMDS, LAB_GHZ = 1024, 2.4


def perm_ceiling_per_s():
    return SIMDS * LAB_GHZ * 1e9 / sum(n * c for _, n, c in PERM_MIX) * 64.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proofs", type=int, default=65536, help="proofs per GPU per step (weak scaling)")
    ap.add_argument("--total-proofs", type=int, default=0,
                    help="proofs of the WHOLE job per step, split contiguously over the ranks (strong scaling; "
                         "BASELINE configs[3] = 1048576)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="oracle sample size (0 = skip, -1 = auto)")
    ap.add_argument("--perm-log2", type=int, default=24, help="Poseidon2 microbench size (log2 states, 0 = skip)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="standard")
    ap.add_argument("--order", choices=["round_robin", "level"], default="round_robin",
                    help="how the job's proofs are ordered: round-robin over the workload's fixtures (the bench line), or level by "
                         "level as the reference's driver produces a recursion chain (examples/multi-proofs/src/main.rs:198-295). "
                         "With --total-proofs the job is cut by rsv_shard_plan (contiguous, balanced by bytes) either way.")
    ap.add_argument("--exchange", choices=["torch", "c"], default="torch",
                    help="the per-step exchange of accept bitmaps: torch.distributed collectives (default: the path the multi-rank "
                         "tests rehearse) or the C-ABI's own (rsv_exchange_*: ncclAllGather + ncclAllReduce issued by the library "
                         "on the verifier's stream, what a Rust host would call)")
    ap.add_argument("--devices", default=None, metavar="D0,D1,...",
                    help="ONE process drives these HIP devices through rsv_multi_verify_batch_dev (one context and host thread "
                         "per entry; a device may repeat), host-assembled bitmap, no collective.  Not combined with --gpus.")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU: K contexts, each verifying its own copy of the batch from its own host "
                         "thread (a step = K batches).  Small batches are latency-bound chains that leave the chip idle; "
                         "independent contexts overlap them.  Not the bench line (default 1).")
    ap.add_argument("--emit-paths", action="store_true",
                    help="also emit every hint output (transcript rows, trace-tree and FRI per-query paths; SURVEY 8f.1) from the "
                         "verifying pass; needs a uniform-shape workload")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="experiments only: a tuning knob of the library (rsv_ctx_set_option, names in rsv.OPTIONS), e.g. "
                         "critical_chain=off.  The bench line is measured with none.")
    ap.add_argument("--witness-proofs", type=int, default=1024,
                    help="side leg: the recursion circuit's witness for this many proofs of the level10 shape (0 = skip; skipped with --perm-log2 0)")
    ap.add_argument("--no-stage-times", dest="stage_times", action="store_false",
                    help="do not record the per-stage HIP events (RSV_OPT_STAGE_TIMES): what a production caller runs; the event "
                         "pairs cost ~8 us per stage, which matters for batches of a few thousand proofs and below (tools/sweep.sh "
                         "measures latency this way).  roofline.kernel_ms / stage_ms are then null.")
    ap.add_argument("--lib", default=None,
                    help="A/B only: another build of the library than csrc/librsv_hip.so (same ABI), e.g. one built from the "
                         "previous commit's kernel sources, measured on the same box")
    ap.add_argument("--rehearsal", action="store_true",
                    help="one-GPU box only: every rank on cuda:0, bitmap exchange over gloo (RCCL refuses two ranks on one "
                         "device).  The driver's runs never pass it: they use one GPU per rank and nccl (= RCCL).")
    ap.add_argument("--no-single-proof", dest="single_proof", action="store_false",
                    help="skip the single-proof latency leg (one proof per call, the reference's own use)")
    ap.add_argument("--emit-flow", action="store_true",
                    help="also emit the PoseidonFlow of every proof's verification circuit (SURVEY 8f.1, second half: 128 B + "
                         "1 B per Poseidon invocation, ~0.7 MB per standard proof) from the verifying pass; any workload")
    args = ap.parse_args()

    # ---- plain `python bench.py --gpus N`: start the N ranks as child processes and wait.  This process loads neither
    # torch nor the HIP library (sharding.py is imported by path; it needs only the standard library and numpy).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import importlib.util
        spec = importlib.util.spec_from_file_location("rsv_sharding_launcher", os.path.join(ROOT, "recursive-stwo_amd", "sharding.py"))
        launcher = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launcher)
        sys.exit(launcher.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    if args.devices is not None:
        if args.gpus != 1:
            raise SystemExit("--devices is the one-process layout: not combined with --gpus")
        return run_one_process(args)

    import rsvload
    rsv = rsvload.load_package(args.lib)
    from recursive_stwo_amd import sharding
    import torch
    import torch.distributed as dist
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    # Rehearsal mode for a one-GPU box (--rehearsal): every rank on cuda:0, the bitmaps exchanged over gloo (RCCL
    # refuses two ranks on one device).  The driver's real runs use nccl (= RCCL).
    rehearsal = args.rehearsal
    if torch.cuda.device_count() < (1 if rehearsal else world_env):
        raise SystemExit("bench.py needs one HIP device per rank: the product has no CPU fallback")
    for kv in args.knob:
        name, value = kv.split("=", 1)
        rsv.set_default_option(name, value if value in rsv.OPTION_VALUES else int(value))
    rank, world, dev_index = sharding.init_rank(torch, dist, rehearsal)
    if rsv.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    dev = torch.device("cuda", dev_index)

    strong = args.total_proofs > 0
    n_total = args.total_proofs if strong else world * args.proofs
    fixtures = WORKLOADS[args.workload]
    # the partition: a job of fixed size is cut by rsv_shard_plan (contiguous, balanced by bytes: every rank computes the
    # same cuts from the job's lengths); per-GPU batches of equal size (weak scaling) are rsv_shard_range's equal counts
    plan = rsv.shard_plan(job_lengths(n_total, fixtures, args.order), world) if strong else None
    first, last = (plan[0][rank], plan[1][rank]) if plan else sharding.shard_range(n_total, rank, world)
    n = last - first
    d_blob, d_offsets, plen, tam, fix_idx = build_batch_on_device(torch, dev, n, first, fixtures, args.order, n_total)
    total_bytes = int(plen.sum())
    # (RCCL prints a version banner on STDOUT when a communicator is first created; this process prints ONE JSON line there)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        sv = sharding.ShardedVerifier(rsv, n_total, rank, world, dev_index, dist, torch, plan=plan, exchange=args.exchange)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if args.stage_times:  # the roofline's kernel time comes from HIP events on the verifier's streams
        try:
            sv.ctx.set_option("stage_times", "on")
        except Exception:
            if not args.lib:  # (an older build under --lib has the clock always on)
                raise
    # the configuration of every proof: the reference's literal for the fixture it was copied from
    fcfg = fixture_configs(rsv, fixtures)
    cfg = sv.ctx.prepare_cfg([fcfg[k] for k in fix_idx], n) if len({rsv._cfg_key(c) for c in fcfg}) > 1 else rsv.PreparedCfg([fcfg[0]])
    hints = None
    if args.emit_paths:
        hdr = np.frombuffer(read_fixture(fixtures[0])[:64], dtype=np.uint32)
        p_nq, p_M = int(hdr[13]), max(int(hdr[0]) + 1, int(hdr[1]) + 2) + int(hdr[11])
        p_inner = p_M - int(hdr[11]) - int(hdr[12]) - 1  # FRI inner layers: folds from log M-blowup down to log_last
        hints = dict(shape=(p_nq, p_M, p_inner),
                     d_transcript=torch.zeros((n, rsv.TRANSCRIPT_WORDS), dtype=torch.int32, device=dev),
                     d_trace_sib=torch.zeros((n, 4, p_nq, p_M, 8), dtype=torch.int32, device=dev),
                     d_trace_pos=torch.zeros((n, 4, p_nq), dtype=torch.int32, device=dev),
                     d_trace_cols=torch.zeros((n, 4, p_nq, 64), dtype=torch.int32, device=dev),
                     d_fri_sib=torch.zeros((n, 1 + p_inner, p_nq, p_M, 8), dtype=torch.int32, device=dev),
                     d_fri_cols=torch.zeros((n, 1 + p_inner, p_nq, 3, 8), dtype=torch.int32, device=dev),
                     d_fri_folded=torch.zeros((n, 3, p_nq, 4), dtype=torch.int32, device=dev))

    if args.emit_flow:
        stride = 0
        for f in fixtures:
            hdr = np.frombuffer(read_fixture(f)[:64], dtype=np.uint32)
            stride = max(stride, rsv.poseidon_flow_count(int(hdr[0]), int(hdr[1]), rsv.PcsConfig(int(hdr[10]), int(hdr[11]), int(hdr[12]), int(hdr[13]))))
        hints = dict(hints or {})
        hints.update(d_flow=torch.empty((n, stride, 32), dtype=torch.int32, device=dev),
                     d_flow_swap=torch.empty((n, stride), dtype=torch.uint8, device=dev),
                     d_flow_count=torch.zeros(n, dtype=torch.int32, device=dev))

    # --inflight K: K - 1 further contexts with their own copy of the batch and of the outputs, driven by K - 1 threads
    extra = []
    if args.inflight > 1:
        from concurrent.futures import ThreadPoolExecutor as _Pool
        for _ in range(args.inflight - 1):
            ctx2 = rsv.Context(dev_index)
            extra.append({"ctx": ctx2, "blob": d_blob.clone(), "acc": torch.zeros(n, dtype=torch.uint8, device=dev),
                          "cfg": ctx2.prepare_cfg([fcfg[k] for k in fix_idx], n) if len({rsv._cfg_key(c) for c in fcfg}) > 1 else rsv.PreparedCfg([fcfg[0]])})
        pool = _Pool(len(extra))
        torch.cuda.synchronize()

    def step():
        futs = [pool.submit(e["ctx"].verify_batch, e["blob"], d_offsets, n, e["acc"], None, e["cfg"]) for e in extra]
        sv.step(d_blob, d_offsets, cfg, hints=hints)
        for f_ in futs:
            f_.result()

    def fence():
        for e in extra:
            e["ctx"].synchronize()
        sv.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    stage_sum = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP-event stage times of this step (reads events already recorded on the verifier's streams)
        if rank == 0 and args.stage_times:
            for k, v in sv.ctx.last_stage_times().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # correctness of what was just timed: every untampered proof accepted, every tampered one rejected
    acc = sv.d_accept[:n].cpu().numpy()
    want = np.ones(n, np.uint8)
    want[tam] = 0
    for e in extra:
        if not np.array_equal(e["acc"].cpu().numpy(), want):
            raise SystemExit("verdict mismatch in an extra in-flight context")
    if not np.array_equal(acc, want):
        raise SystemExit(f"rank {rank}: verdict mismatch: {int((acc != want).sum())} proofs differ from the expected accept map")
    # every rank now holds the whole job's accept bitmap and count: check them against the global tamper rule
    job = sv.exchange.assemble()
    job_want = (np.arange(n_total) % 17 != 5).astype(np.uint8)
    if not np.array_equal(job, job_want):
        raise SystemExit(f"rank {rank}: gathered bitmap differs from the expected accept map of the job")
    if sv.exchange.total_accepted() != int(job_want.sum()):
        raise SystemExit("accept count (all-reduce) mismatch")

    # what the process group looked like from inside (a SCALE record should prove that N ranks on N devices took part)
    def dev_id():
        pr = torch.cuda.get_device_properties(dev_index)
        return {"rank": rank, "device_index": dev_index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
                "pci_bus_id": getattr(pr, "pci_bus_id", None), "pid": os.getpid()}
    group = {"world_size": 1, "backend": None, "devices": [dev_id()]}
    if dist.is_initialized():
        devs = [None] * dist.get_world_size()
        dist.all_gather_object(devs, dev_id())
        ver = None
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            pass
        group = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver, "devices": devs}

    # bytes of every rank's shard (what the partition balances)
    shard_bytes = [total_bytes]
    if dist.is_initialized():
        sb = [None] * dist.get_world_size()
        dist.all_gather_object(sb, total_bytes)
        shard_bytes = sb
    if args.exchange == "c":
        v = rsv.lib.rsv_exchange_rccl_version()
        exchange_backend = f"rsv_exchange (rccl {v // 10000}.{v // 100 % 100}.{v % 100})"
    else:
        exchange_backend = f"torch.distributed ({group['backend']})" if group["backend"] else "torch tensors, no process group (1 rank)"

    if rank != 0:
        if args.exchange == "c":
            sv.exchange.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    ctx = sv.ctx
    ms_per_step = dt / args.steps * 1e3
    value = n_total * args.inflight * args.steps / dt
    stage_avg = {k: v / args.steps for k, v in stage_sum.items()}
    if stage_avg:
        dom = max((k for k in stage_avg if k.endswith("merkle")), key=lambda k: stage_avg[k])
        dom_ms = stage_avg[dom]
    else:  # --no-stage-times: no kernel time, so achieved / frac / valu_issue_frac are not computed (0 / null)
        dom, dom_ms = "pair_merkle", 0.0
    algo_bytes = total_bytes + n  # proof bytes read once + 1 accept byte per proof
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: PMC bytes measured by tools/profile.sh on THESE kernel sources, else null.
    traffic, traffic_note = None, "traffic: no PMC profile of these kernel sources under profiles/ (tools/profile.sh writes pmc_latest.json)"
    valu_issue_frac = None
    valu_issue_frac_solo = kernel_solo_ms = None  # the profiled launch running ALONE (counter pass): counters and time of one run
    pipeline_valu_issue_frac = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            prof = json.load(f)
        if prof.get("kernel_sources_sha") != kernel_sources_sha():
            traffic_note = (f"traffic: profiles/pmc_latest.json was measured on other kernel sources "
                            f"({prof.get('kernel_sources_sha')}), not reported")
        elif args.workload != "standard" or n != prof.get("proofs"):
            traffic_note = "traffic: profiles/pmc_latest.json holds the default workload only, not reported for this one"
        else:
            pk = prof["kernels"]["k_" + dom]
            traffic = float(pk["hbm_bytes_corrected"])
            # raw VALU issue of the dominant kernel: wave-level VALU instructions of the profiled launch over THIS run's
            # kernel time, against one wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz (the nominal rate)
            if pk.get("SQ_INSTS_VALU") and dom_ms > 0:
                valu_issue_frac = float(pk["SQ_INSTS_VALU"]) / (dom_ms * 1e-3) / (SIMDS * LAB_GHZ * 1e9 / 2.0)
            valu_issue_frac_solo, kernel_solo_ms = pk.get("valu_issue_frac_solo"), pk.get("solo_ms")
            # the whole step: every pipeline kernel's VALU instructions over the step's wall time (kernels overlap, so a
            # single kernel's span also holds the others' instructions; this figure does not depend on the attribution)
            if prof.get("pipeline_valu_insts"):
                pipeline_valu_issue_frac = float(prof["pipeline_valu_insts"]) / (dt / args.steps) / (SIMDS * LAB_GHZ * 1e9 / 2.0)
            traffic_note = (f"traffic: HBM bytes of k_{dom} per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                            f"command on these kernel sources ({prof['tag']}), FETCH_SIZE doubled per MI355X_MICROARCH.md")
    except (OSError, KeyError, ValueError):
        pass
    pipeline_gbps = algo_bytes / (ms_per_step * 1e-3) / 1e9
    # `bound` says what really binds the dominant kernel: integer VALU issue (99.8 % of its instructions are the
    # permutation).  achieved / peak / frac stay the tier's nominal HBM figure (algorithmic bytes over kernel time against
    # 8 TB/s), as SURVEY 8d defines them; valu_issue_frac is the binding one.
    roofline = {"bound": "valu", "nominal_bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "valu_issue_frac": valu_issue_frac,
                "valu_issue_frac_solo": valu_issue_frac_solo, "kernel_solo_ms": kernel_solo_ms,
                "pipeline_valu_issue_frac": pipeline_valu_issue_frac,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": dom_ms if stage_avg else None,
                "pipeline_ms": ms_per_step, "pipeline_GBps": pipeline_gbps, "pipeline_frac": pipeline_gbps / HBM_PEAK_GBPS,
                "stage_ms": stage_avg or None,
                "note": "31-bit modular integer hashing: VALU-issue bound, not HBM bound (SURVEY §8d); see valu. "
                        "frac charges the whole proof to the dominant kernel (SURVEY §8d's numerator); pipeline_frac "
                        "is the same bytes over the wall time of a step on rank 0.  valu_issue_frac = SQ_INSTS_VALU of the "
                        "profiled launch / kernel_ms / (1 024 SIMDs x 2.4 GHz / 2); valu_issue_frac_solo = the same counter over kernel_solo_ms, "
                        "the launch's duration in the counter pass, where it runs alone (kernel_ms is a span beside the other stream's "
                        "kernels).  kernel_ms / stage_ms are HIP-event "
                        "spans on the verifier's streams; side-stream stages overlap the main stream, so stage_ms do not "
                        "add up to pipeline_ms, and a span includes the time a launch waits for room on the chip: "
                        "'cap_top(span)' is ~0.5 ms of work that sits underneath the FRI trees for most of their duration.  " + traffic_note}

    # What the reference's own use looks like: ONE proof per call (examples/single-proof/src/main.rs:23-82).  Latency of
    # rsv_verify_batch_dev + rsv_ctx_synchronize for one proof resident in HBM, median of 30 calls: a chain of dependent
    # kernels (233 transcript permutations, then ~25 per Merkle path), not a throughput figure.
    single = None
    if world == 1 and args.single_proof:
        sp = read_fixture(fixtures[0])
        d_sp = torch.from_numpy(np.frombuffer(sp, dtype=np.uint8).copy()).to(dev)
        d_so = torch.tensor([0, len(sp)], dtype=torch.int64, device=dev)
        d_sa = torch.zeros(1, dtype=torch.uint8, device=dev)
        scfg = rsv.PreparedCfg([fcfg[0]])
        if not args.lib:
            ctx.set_option("stage_times", "off")  # as a caller runs it: the stage clock is a diagnostic (~8 us per stage)
        lat = []
        for k in range(35):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.verify_batch(d_sp, d_so, 1, d_sa, None, scfg)
            ctx.synchronize()
            if k >= 5:
                lat.append(time.perf_counter() - t1)
        if int(d_sa.item()) != 1:
            raise SystemExit("single proof: a genuine fixture was not accepted")
        lat.sort()
        single = {"latency_ms": lat[len(lat) // 2] * 1e3, "min_ms": lat[0] * 1e3, "calls": len(lat), "fixture": fixtures[0],
                  "note": "one proof per call, resident in HBM: rsv_verify_batch_dev + rsv_ctx_synchronize, median wall time"}

    # Poseidon2 microbench (second metric of BASELINE.json): 2^k states resident in HBM, 128 B per permutation
    valu = None
    if args.perm_log2 > 0:
        m = 1 << args.perm_log2
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        d_in = torch.randint(0, 0x7FFFFFFF, (m, 16), dtype=torch.int32, device=dev, generator=gen)
        d_out = torch.empty_like(d_in)
        for _ in range(3):  # untimed: module load, clocks back up after the host-side pause above
            ctx.poseidon2_permute(d_in, d_out)
        ctx.synchronize()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            ctx.poseidon2_permute(d_in, d_out)
        ctx.synchronize()
        pdt = (time.perf_counter() - t1) / reps
        perms_per_s = m / pdt
        ceil = perm_ceiling_per_s()
        valu = {"poseidon2_perms_per_s": perms_per_s, "perm_GBps": m * 128 / pdt / 1e9,
                "perm_hbm_frac": m * 128 / pdt / 1e9 / HBM_PEAK_GBPS, "states": m,
                "ceiling_perms_per_s": ceil, "frac_of_ceiling": perms_per_s / ceil,
                "ceiling_model": "sum over instruction classes of count x measured issue cost (tools/valu_lab.hip, 4 waves/SIMD) = "
                                 f"{sum(n * c for _, n, c in PERM_MIX):.0f} cycles-at-2.4-GHz per 64 permutations per SIMD, {SIMDS} SIMDs "
                                 "(tools/perm_ceiling.py, DESIGN §4).  The lab's cycles are wall time x 2.4 GHz AND real shader cycles: under "
                                 "its dense VALU load GRBM_GUI_ACTIVE holds 2.35-2.40 GHz (round 5: profiles/r5_valu_lab_grbm_clock.txt), so the "
                                 "fast class's 2.5 cycles per wave-instruction against the nominal 2 are issue overhead, not a sagging clock"}
        del d_in, d_out

    # SURVEY §8f.4 widening: gate values of the emulated Poseidon2, HBM-write bound (65 B in, 6 656 B out per permutation)
    emulated = None
    if args.perm_log2 > 0 and world == 1:
        m = 1 << min(args.perm_log2, 20)
        gen = torch.Generator(device=dev)
        gen.manual_seed(2)
        d_l = torch.randint(0, 0x7FFFFFFF, (m, 8), dtype=torch.int32, device=dev, generator=gen)
        d_r = torch.randint(0, 0x7FFFFFFF, (m, 8), dtype=torch.int32, device=dev, generator=gen)
        d_s = torch.randint(0, 3, (m,), dtype=torch.uint8, device=dev, generator=gen)
        d_rows = torch.empty((m, rsv.EMU_STRIDE, 4), dtype=torch.int32, device=dev)
        for _ in range(2):
            ctx.poseidon2_emulated(d_l, d_r, d_s, d_rows)
        ctx.synchronize()
        t1 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            ctx.poseidon2_emulated(d_l, d_r, d_s, d_rows)
        ctx.synchronize()
        edt = (time.perf_counter() - t1) / reps
        ebytes = 65 + rsv.EMU_STRIDE * 16
        emulated = {"perms_per_s": m / edt, "perms": m, "algorithmic_bytes_per_perm": ebytes, "bound": "hbm",
                    "achieved_GBps": m * ebytes / edt / 1e9, "peak_GBps": HBM_PEAK_GBPS,
                    "frac": m * ebytes / edt / 1e9 / HBM_PEAK_GBPS}
        del d_l, d_r, d_s, d_rows

    # SURVEY §8f.1 widening: the recursion circuit's `variables` for a batch (rsv_witness_eval_dev): 1 024 proofs of the
    # level10 shape, program built here from the GPU's own hints of the fixture; HBM streaming
    witness = None
    if args.perm_log2 > 0 and args.witness_proofs > 0 and world == 1 and rank == 0:
        wname = "level10-1.bin"
        wproof = read_fixture(wname)
        wcfg = fixture_configs(rsv, [wname])[0]
        t1 = time.perf_counter()
        wp = rsv.WitnessProgram.build(wproof, wcfg, device=dev_index)
        wbuild = time.perf_counter() - t1
        wn = args.witness_proofs
        wblob, woff = rsv.pack([wproof] * wn)
        d_wblob, d_woff = torch.from_numpy(wblob.copy()).to(dev), torch.from_numpy(woff.astype(np.int64)).to(dev)
        d_wvars = torch.empty((wn, wp.n_vars, 4), dtype=torch.int32, device=dev)
        d_wacc = torch.zeros(wn, dtype=torch.uint8, device=dev)
        ctx.witness(wp, d_wblob, d_woff, wn, d_wvars, d_wacc)
        ctx.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.witness(wp, d_wblob, d_woff, wn, d_wvars, d_wacc)
        ctx.synchronize()
        wdt = (time.perf_counter() - t1) / reps
        if not bool(d_wacc.all().item()):
            raise SystemExit("witness leg: a genuine proof was not accepted")
        witness = {"proofs_per_s": wn / wdt, "ms": wdt * 1e3, "proofs": wn, "fixture": wname, "variables_per_proof": wp.n_vars,
                   "levels": wp.n_levels, "output_GB": wn * wp.n_vars * 16 / 1e9, "program_build_s": wbuild,
                   "includes": "the verifying pass with the hint outputs the program reads, the program (batches up to 1 024 proofs of a short program: one launch, a workgroup per four proofs; otherwise the wide head a launch per level and everything behind it in one), the transpose "
                               "(split and roofline: tools/bench_witness.py, profiles/r3_witness_*.json)"}
        wp.close()
        del d_wblob, d_wvars

    cpu = None
    host_path = None
    sample = args.cpu_sample
    if sample != 0 and world == 1:  # the CPU baseline is a single-GPU-run figure (rank 0 at N = 1 only)
        if sample < 0:
            sample = 10240  # the legs of cpu_baseline() and the host-path sample are cut from it
        n_s = min(sample, n)
        end = int(d_offsets[n_s].item())
        blob_host = d_blob[:end].cpu().numpy()
        rows = [rsv._cfg_key(c) for c in fcfg]
        table = sorted(set(rows))
        of = np.array([table.index(rows[k]) for k in fix_idx[:n_s]], np.uint8)
        offs_host = d_offsets[:n_s + 1].cpu().numpy()
        # PCIe-inclusive rate (never `value`): the same sample starting in HOST memory, one buffer per proof as the
        # reference's callers hold them, through rsv_verify_batch_host (gather -> pinned -> DMA -> verify, pipelined)
        n_h = min(10240, n_s)  # the host-path sample of rounds 1-3 (1.2 GB); tools/host_path_bench.py sweeps sizes
        views = [blob_host[int(offs_host[i]):int(offs_host[i + 1])] for i in range(n_h)]
        hcfg = [fcfg[k] for k in fix_idx[:n_h]] if len(set(rows)) > 1 else fcfg[0]
        hb = rsv.HostBatch(views)  # the pointer table a Rust caller's Vec<Vec<u8>> already is
        ctx.verify_batch_host(hb, hcfg)  # untimed: the context's pinned staging ring is allocated on first use (~1 GB/s)

        def median_of_5(batch):
            times, acc_ = [], None
            for _ in range(5):
                th = time.perf_counter()
                acc_, _r = ctx.verify_batch_host(batch, hcfg)
                times.append(time.perf_counter() - th)
            times.sort()
            return times[2], times[0], times[-1], acc_
        hdt, hmin, hmax, hacc = median_of_5(hb)
        if not np.array_equal(hacc, want[:n_h]):
            raise SystemExit("host path: verdict mismatch")
        host_path = {"value": n_h / hdt, "unit": "proofs/s", "GBps": hb.bytes / hdt / 1e9, "proofs": n_h, "calls": 5,
                     "GBps_min_max": [hb.bytes / hmax / 1e9, hb.bytes / hmin / 1e9],
                     "note": "rsv_verify_batch_host on the first 10 240 proofs of the batch, MEDIAN of 5 calls behind an untimed first one: "
                             "proofs start in pageable host memory, one buffer each; includes the gather into pinned memory, the PCIe "
                             "upload, the verdict download.  Never `value`."}
        # the caller that cooperates: the same proofs read back to back into the library's pinned arena (rsv_host_alloc):
        # the DMA engine uploads every chunk from where it is, no gather copy
        arena = rsv.HostArena(hb.bytes + 4096)
        ha = arena.pack(views)
        ctx.verify_batch_host(ha, hcfg)
        adt, amin, amax, aacc = median_of_5(ha)
        if not np.array_equal(aacc, want[:n_h]):
            raise SystemExit("host path (pinned arena): verdict mismatch")
        host_path["pinned_arena"] = {"value": n_h / adt, "unit": "proofs/s", "GBps": ha.bytes / adt / 1e9, "calls": 5,
                                     "GBps_min_max": [ha.bytes / amax / 1e9, ha.bytes / amin / 1e9],
                                     "note": "the same proofs held back to back in rsv_host_alloc memory: uploaded from where they are"}
        del ha
        arena.close()
        cpu = cpu_baseline(blob_host, offs_host, n_s, table, of, fixtures)
        if witness is not None:
            # the witness leg's CPU baseline: the oracle's restatement of the circuit (Python integers) on ONE proof, one thread
            from tests import oracle_binding as ob
            from oracle import recursion_circuit as rc
            t1 = time.perf_counter()
            c_or, _, _ = rc.build_circuit(read_fixture(witness["fixture"]), ob)
            odt = time.perf_counter() - t1
            if len(c_or.variables) != witness["variables_per_proof"]:
                raise SystemExit("witness leg: the oracle's circuit has another number of variables")
            witness["cpu_baseline"] = {"value": 1.0 / odt, "unit": "proofs/s", "cores": 1, "kind": "port",
                                       "sample": "one proof: oracle/recursion_circuit (the gadgets run in Python integers on the C "
                                                 "oracle's hints) — a restatement for checking, not a tuned implementation"}
        if valu is not None and cpu.get("perms_per_proof"):
            # useful permutations (the oracle's batched-walk count) the whole pipeline retires per second and GPU, against
            # the instruction-cost ceiling of the bare permutation: the efficiency figure of this VALU-bound path
            valu["pipeline_useful_perms_per_s_per_gpu"] = cpu["perms_per_proof"] * value / world
            valu["pipeline_frac_of_ceiling"] = valu["pipeline_useful_perms_per_s_per_gpu"] / valu["ceiling_perms_per_s"]

    if strong:
        wl = (f"{'BASELINE configs[3]' if args.workload == 'standard' else args.workload}: {n_total} proofs per step split "
              f"contiguously over {world} GPU(s), round-robin over {fixtures}")
    else:
        wl = (f"{'BASELINE configs[2]' if args.workload == 'standard' else args.workload}: {n} proofs/GPU round-robin over {fixtures}")
    line = {
        "metric": "recursive proofs verified/sec", "value": value, "unit": "proofs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32 (M31 modular integers)", "data": "synthetic",
        "config": {"workload": wl + ", i%17==5 tampered (SURVEY §8d), full verify under the reference's PcsConfig literals; "
                                    "bit-exact accept map of the whole job checked on every rank",
                   "proofs_per_step": n_total * args.inflight, "proofs_rank0": n, "bytes_rank0": total_bytes, "parallelism": f"shard{world}",
                   "batches_in_flight": args.inflight, "knobs": args.knob,
                   "hint_outputs": sorted(k for k in (hints or {}) if k.startswith("d_")),
                   "flow_bytes_per_step_rank0": (int(hints["d_flow"].numel()) * 4 + int(hints["d_flow_swap"].numel())) if args.emit_flow else 0,
                   "order": args.order,
                   "partition": ("rsv_shard_plan: contiguous, balanced by bytes" if plan else "rsv_shard_range: contiguous, equal counts"),
                   "shard_bytes": shard_bytes,
                   "exchange": dict(group, backend=exchange_backend,
                                    collectives="all_gather(accept bitmap) + all_reduce(count) per step, " + (
                       "gloo (rehearsal)" if rehearsal else ("nccl/RCCL" if world > 1 else "none (1 rank)")))},
        "roofline": roofline, "cpu_baseline": cpu, "host_path": host_path, "valu": valu, "emulated_poseidon2": emulated,
        "recursion_circuit_witness": witness, "single_proof": single,
    }
    print(json.dumps(line), flush=True)
    if args.exchange == "c":
        sv.exchange.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def run_one_process(args):
    """`--devices D0,D1,...`: the one-process layout of the C-ABI (rsv_multi_verify_batch_dev) — what the reference's own
    driver is (one process walks the whole job).  One context and one host thread per entry, every shard resident on its
    device, the job's bitmap and count assembled on the host: no collective.  Prints the standard line (its `roofline`
    carries the whole-step figure only: the per-kernel clock belongs to the per-rank path)."""
    import rsvload
    rsv = rsvload.load_package(args.lib)
    import torch
    devices = [int(x) for x in args.devices.split(",") if x != ""]
    world = len(devices)
    if world < 1 or rsv.device_count() <= max(devices):
        raise SystemExit("bench.py --devices needs every named HIP device: the product has no CPU fallback")
    for kv in args.knob:
        name, value = kv.split("=", 1)
        rsv.set_default_option(name, value if value in rsv.OPTION_VALUES else int(value))
    strong = args.total_proofs > 0
    n_total = args.total_proofs if strong else world * args.proofs
    fixtures = WORKLOADS[args.workload]
    fcfg = fixture_configs(rsv, fixtures)
    rows = [rsv._cfg_key(c) for c in fcfg]
    table = sorted(set(rows))
    if strong:
        lo, hi = rsv.shard_plan(job_lengths(n_total, fixtures, args.order), world)
    else:
        lo, hi = zip(*[rsv.shard_range(n_total, r, world) for r in range(world)])
    shards, shard_bytes, keep = [], [], []
    for r, d in enumerate(devices):
        dev = torch.device("cuda", d)
        n = hi[r] - lo[r]
        d_blob, d_offsets, plen, _, fix_idx = build_batch_on_device(torch, dev, n, lo[r], fixtures, args.order, n_total)
        d_of = torch.from_numpy(np.array([table.index(rows[k]) for k in fix_idx], np.uint8)).to(dev) if len(table) > 1 else None
        d_acc = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        shards.append({"d_blob": d_blob, "d_offsets": d_offsets, "n": n, "d_cfg_of": d_of, "d_accept": d_acc})
        shard_bytes.append(int(plen.sum()))
        keep.append((d_blob, d_offsets, d_of, d_acc))
    for d in set(devices):
        torch.cuda.synchronize(d)
    mc = rsv.MultiContext(devices)
    cfg_table = [rsv.PcsConfig(*k) for k in table]
    for _ in range(args.warmup):
        mc.verify_batch_dev(shards, cfg_table)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bitmap, count = mc.verify_batch_dev(shards, cfg_table)  # returns when every context is done and the bitmap assembled
    dt = time.perf_counter() - t0
    job_want = (np.arange(n_total) % 17 != 5).astype(np.uint8)
    job = np.unpackbits(bitmap.view(np.uint8), bitorder="little")[:n_total]
    if not np.array_equal(job, job_want) or count != int(job_want.sum()):
        raise SystemExit("one-process path: the job's bitmap or count differs from the expected accept map")
    for r, sh in enumerate(shards):
        if not np.array_equal(sh["d_accept"][: sh["n"]].cpu().numpy(), job_want[lo[r]:hi[r]]):
            raise SystemExit(f"one-process path: shard {r}'s accept bytes differ from the expected accept map")
    ms_per_step = dt / args.steps * 1e3
    algo_bytes = sum(shard_bytes) + n_total
    gbps = algo_bytes / (ms_per_step * 1e-3) / 1e9
    names = []
    for d in devices:
        pr = torch.cuda.get_device_properties(d)
        names.append({"device_index": d, "name": pr.name, "uuid": str(getattr(pr, "uuid", ""))})
    line = {
        "metric": "recursive proofs verified/sec", "value": n_total * args.steps / dt, "unit": "proofs/s", "n_gpus": len(set(devices)),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32 (M31 modular integers)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_total} proofs per step over {world} context(s) of ONE process, {args.order} over {fixtures}, "
                               "i%17==5 tampered (SURVEY §8d), full verify; the job's bitmap, count and every shard's accept bytes checked",
                   "proofs_per_step": n_total, "parallelism": f"one process, contexts on devices {devices}", "knobs": args.knob, "order": args.order,
                   "partition": "rsv_shard_plan: contiguous, balanced by bytes" if strong else "rsv_shard_range: contiguous, equal counts",
                   "shard_bytes": shard_bytes,
                   "exchange": {"backend": "rsv_multi (host assembly)", "world_size": world, "devices": names,
                                "collectives": "none: one process owns every shard; bitmap slices come back over PCIe (n / 8 bytes)"}},
        "roofline": {"bound": "valu", "nominal_bound": "hbm", "kernel": None, "achieved": gbps, "peak": HBM_PEAK_GBPS * len(set(devices)), "unit": "GB/s",
                     "frac": gbps / (HBM_PEAK_GBPS * len(set(devices))), "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                     "pipeline_ms": ms_per_step,
                     "note": "whole-step figure (algorithmic bytes over the wall time of a blocking rsv_multi_verify_batch_dev call, "
                             "against the named devices' HBM peak); the per-kernel roofline is the default path's"},
        "cpu_baseline": None,
    }
    print(json.dumps(line), flush=True)
    mc.close()


if __name__ == "__main__":
    main()
