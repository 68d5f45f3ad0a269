#!/usr/bin/env python3
"""bench.py — recursive proofs verified / s on MI355X (BASELINE.json metric).

A "step" is one pass of the verify pipeline over one synthetic batch that is already resident in HBM:
BASELINE configs[2], 65 536 proofs built round-robin from the reference's four standard-config fixtures
(recursive_proof_16_15, level3-1, level6-1, level7-1) with the seeded tamper rule of SURVEY §8d
(proof i with i % 17 == 5 gets one flipped bit).  With --gpus N every rank verifies its own
65 536-proof shard of an N x 65 536 batch (weak scaling) and the accept bitmaps are exchanged with one
all_gather per step (RCCL).  Rank 0 prints ONE JSON line.

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

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


def build_batch_on_device(torch, dev, n_proofs, first_index, fixtures=FIXTURES):
    """Returns (d_blob uint8, d_offsets int64[n+1], lengths, tamper byte offsets) for global proof
    indices [first_index, first_index + n_proofs)."""
    proofs = [read_fixture(f) for f in fixtures]
    lens = np.array([len(p) for p in proofs], dtype=np.int64)
    idx = (np.arange(n_proofs, dtype=np.int64) + first_index) % len(proofs)
    plen = lens[idx]
    offsets = np.zeros(n_proofs + 1, dtype=np.int64)
    np.cumsum(plen, out=offsets[1:])
    total = int(offsets[-1])
    # one period (4 proofs, aligned to the round-robin phase) tiled across the batch, built in HBM
    phase = int(first_index % len(proofs))
    order = [(phase + k) % len(proofs) for k in range(len(proofs))]
    period = np.frombuffer(b"".join(proofs[k] for k in order), dtype=np.uint8)
    d_period = torch.from_numpy(period.copy()).to(dev)
    reps = (n_proofs + len(proofs) - 1) // len(proofs)
    d_blob = d_period.repeat(reps)[:total].contiguous()
    # seeded tampering
    tam = [i for i in range(n_proofs) if (first_index + i) % 17 == 5]
    pos = np.array([offsets[i] + 60 + splitmix64(0xC0FFEE, first_index + i) % (int(plen[i]) - 68) for i in tam],
                   dtype=np.int64)
    if len(pos):
        d_pos = torch.from_numpy(pos).to(dev)
        d_blob[d_pos] = d_blob[d_pos] ^ 1
    d_offsets = torch.from_numpy(offsets).to(dev)
    return d_blob, d_offsets, plen, np.array(tam, dtype=np.int64)


def cpu_baseline(blob_host, offsets, n_sample, fixtures=None):
    """The C oracle (a port of the reference algorithm, not the Rust binary) on the host cores."""
    from tests import oracle_binding as ob
    threads = max(1, min(os.cpu_count() or 1, 16))
    n_sample = min(n_sample, len(offsets) - 1)
    bounds = np.linspace(0, n_sample, threads + 1).astype(int)
    pi = ob.make_inputs(ob.STANDARD_INPUTS)

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        if hi <= lo:
            return 0
        offs = np.ascontiguousarray(offsets[lo:hi + 1], dtype=np.uint64)
        acc = np.zeros(hi - lo, np.uint8)
        rc = ob.lib.rsvo_verify_batch(blob_host.ctypes.data_as(ob._u8p), offs.ctypes.data_as(ob._u64p), hi - lo, None,
                                      pi, 3, acc.ctypes.data_as(ob._u8p), None)
        assert rc == 0
        return int(acc.sum())

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        accepted = sum(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    # Poseidon2 permutations the oracle's batched walk spends on the genuine fixtures of this workload (untimed): the
    # unit of useful work of this path, used for the in-situ permutation rate reported next to the microbenchmark
    perms = [ob.perm_count(read_fixture(f)) for f in fixtures] if fixtures else []
    return {"value": n_sample / dt, "unit": "proofs/s", "cores": threads, "kind": "port",
            "sample": f"first {n_sample} proofs of the rank-0 batch ({accepted} accepted), C oracle, {threads} threads",
            "perms_per_proof": (sum(perms) / len(perms)) if perms else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proofs", type=int, default=65536, help="proofs per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="oracle sample size (0 = skip, -1 = auto)")
    ap.add_argument("--perm-log2", type=int, default=24, help="Poseidon2 microbench size (log2 states, 0 = skip)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="standard")
    ap.add_argument("--emit-paths", action="store_true",
                    help="also emit every hint output (transcript rows, trace-tree and FRI per-query paths; SURVEY 8f.1) from the "
                         "verifying pass; needs a uniform-shape workload")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rsvload
    rsv = rsvload.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (see module docstring)")
    if not torch.cuda.is_available() or rsv.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    # Rehearsal mode for a one-GPU box: RSV_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges the
    # bitmaps over gloo (RCCL refuses two ranks on one device).  The driver's real runs use nccl (= RCCL).
    rehearsal = os.environ.get("RSV_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n = args.proofs
    first = rank * n
    fixtures = WORKLOADS[args.workload]
    d_blob, d_offsets, plen, tam = build_batch_on_device(torch, dev, n, first, fixtures)
    total_bytes = int(plen.sum())
    d_accept = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_reason = torch.zeros(n, dtype=torch.uint8, device=dev)
    n_words = (n + 31) // 32
    d_bitmap = torch.zeros(n_words, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    d_all = torch.zeros(world * n_words, dtype=torch.int32, device=dev) if world > 1 else None
    ctx = rsv.Context(dev_index)
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

    def step():
        if hints:
            ctx.verify_hints(d_blob, d_offsets, n, d_accept, d_reason, **hints)
        else:
            ctx.verify_batch(d_blob, d_offsets, n, d_accept, d_reason)
        ctx.accept_bitmap(d_accept, n, d_bitmap, d_count)
        if world > 1:
            ctx.synchronize()  # bitmap produced on the verifier's stream, exchanged on torch's
            if rehearsal:
                h_all = torch.zeros(world * n_words, dtype=torch.int32)
                dist.all_gather_into_tensor(h_all, d_bitmap.cpu())
                d_all.copy_(h_all)
            else:
                dist.all_gather_into_tensor(d_all, d_bitmap)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
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
        # HIP-event stage times of this step (reads events already recorded on the stream)
        if rank == 0:
            for k, v in ctx.last_stage_times().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # correctness of what was just timed: every untampered proof accepted, every tampered one rejected
    acc = d_accept.cpu().numpy()
    want = np.ones(n, np.uint8)
    want[tam] = 0
    if not np.array_equal(acc, want):
        raise SystemExit(f"rank {rank}: verdict mismatch: {int((acc != want).sum())} proofs differ from the expected accept map")
    if int(d_count.item()) != int(want.sum()):
        raise SystemExit("accept popcount mismatch")
    if world > 1:
        # every rank now holds the whole job's accept bitmap: check it against the global tamper rule
        allbits = np.unpackbits(d_all.cpu().numpy().view(np.uint8), bitorder="little").reshape(world, -1)[:, :n]
        for r in range(world):
            exp = ((np.arange(n) + r * n) % 17 != 5).astype(np.uint8)
            if not np.array_equal(allbits[r], exp):
                raise SystemExit(f"rank {rank}: gathered bitmap of rank {r} differs from the expected accept map")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt
    stage_avg = {k: v / args.steps for k, v in stage_sum.items()}
    dom = max((k for k in stage_avg if k.endswith("merkle")), key=lambda k: stage_avg[k])
    dom_ms = stage_avg[dom]
    algo_bytes = total_bytes + n  # proof bytes read once + 1 accept byte per proof
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # measured HBM traffic of the dominant kernel from profiles/ (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    # passes of this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md), scaled to this batch size
    traffic = None
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", "r1_final_pmc_summary.csv")) as f:
            rows = list(csv.reader(line for line in f if not line.startswith("#")))
        hdr = rows[0]
        for r in rows[1:]:
            if dom in r[0]:
                traffic = float(r[hdr.index("hbm_bytes_corrected")]) * (n / 65536.0)
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": dom_ms,
                "pipeline_ms": ms_per_step, "pipeline_GBps": algo_bytes / (ms_per_step * 1e-3) / 1e9,
                "stage_ms": stage_avg,
                "note": "31-bit modular integer hashing: VALU-issue bound, not HBM bound (SURVEY \u00a78d); see valu. "
                        "kernel_ms / stage_ms are HIP-event times on the verifier's streams; side-stream stages "
                        "overlap the main stream, so stage_ms do not add up to pipeline_ms (wall time per step). "
                        "traffic: PMC bytes of the dominant kernel from profiles/r1_final_pmc_summary.csv"}

    # Poseidon2 microbench (second metric of BASELINE.json): 2^k states resident in HBM, 128 B per permutation
    valu = None
    if args.perm_log2 > 0:
        m = 1 << args.perm_log2
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        d_in = torch.randint(0, 0x7FFFFFFF, (m, 16), dtype=torch.int32, device=dev, generator=gen)
        d_out = torch.empty_like(d_in)
        ctx.poseidon2_permute(d_in, d_out)
        ctx.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.poseidon2_permute(d_in, d_out)
        ctx.synchronize()
        pdt = (time.perf_counter() - t1) / reps
        perms_per_s = m / pdt
        valu = {"poseidon2_perms_per_s": perms_per_s, "perm_GBps": m * 128 / pdt / 1e9,
                "perm_hbm_frac": m * 128 / pdt / 1e9 / HBM_PEAK_GBPS, "states": m}
        del d_in, d_out

    cpu = None
    sample = args.cpu_sample
    if sample != 0:
        threads = max(1, min(os.cpu_count() or 1, 16))
        if sample < 0:
            sample = 640 * threads  # ~5 s per thread at ~8 ms per proof
        n_s = min(sample, n)
        end = int(d_offsets[n_s].item())
        blob_host = d_blob[:end].cpu().numpy()
        cpu = cpu_baseline(blob_host, d_offsets[:n_s + 1].cpu().numpy(), n_s, fixtures)
        if valu is not None and cpu.get("perms_per_proof"):
            # useful permutations (the oracle's batched-walk count) the pipeline retires per second, against the bare
            # permutation kernel measured above: the efficiency figure of this VALU-bound path
            valu["pipeline_useful_perms_per_s"] = cpu["perms_per_proof"] * value
            valu["pipeline_frac_of_perm_kernel"] = valu["pipeline_useful_perms_per_s"] / (world * valu["poseidon2_perms_per_s"])

    line = {
        "metric": "recursive proofs verified/sec", "value": value, "unit": "proofs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32 (M31 modular integers)", "data": "synthetic",
        "config": {"workload": f"{'BASELINE configs[2]' if args.workload == 'standard' else args.workload}: {n} proofs/GPU round-robin over {fixtures}, i%17==5 tampered "
                               f"(SURVEY §8d), full verify; bit-exact accept map checked",
                   "proofs_per_gpu": n, "bytes_per_gpu": total_bytes, "parallelism": f"shard{world}"},
        "roofline": roofline, "cpu_baseline": cpu, "valu": valu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
