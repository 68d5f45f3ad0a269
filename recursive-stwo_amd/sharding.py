"""Multi-GPU host logic: one process per GPU, proofs sharded by contiguous index range, ONE exchange per batch.

Proofs are independent units, so there is no data-path collective: rank r verifies the contiguous
shard [lo_r, hi_r) of the job on its own GPU, packs its verdicts into a bitmap (rsv_accept_bitmap_dev) and the
ranks exchange
    * one all-gather of the per-rank bitmaps (2 KiB per rank for 65 536 proofs: latency-bound, so ONE collective
      of fixed-size slices rather than a ring of variable ones), and
    * one all-reduce(sum) of the accept counts as a cross-check,
over RCCL / xGMI (`torch.distributed` backend "nccl" IS RCCL on ROCm).  In the CPU tests and in the one-GPU
rehearsal the same code runs over "gloo" (tensors staged through host memory, which gloo needs).

`ShardedVerifier.step()` is the per-batch unit `bench.py` times and `tests/test_sharding.py` /
`tests/test_multi_gpu.py` check; `launch_ranks()` starts the N rank processes of a one-node job for a caller
that was itself started as a plain process (`python bench.py --gpus N`).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np


# ----------------------------------------------------------------------------------------------- partition
def shard_range(n_total: int, rank: int, world: int):
    """Contiguous [lo, hi) of proof indices owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_plan(lens, world: int):
    """Work-balanced contiguous cuts (rsv_shard_plan, restated): rank r owns [lo[r], hi[r]); the cut between ranks r - 1 and r
    is the proof boundary nearest to r / world of the job's bytes — bytes are the work (the verifier is permutation-bound
    and every shape of the reference costs 21-26.5 proof bytes per permutation).  For a job that arrives ordered by level
    (examples/multi-proofs/src/main.rs:198-295) equal-count shards differ 4.5 x in bytes.  tests/test_sharding.py holds this
    restatement against the C entry point."""
    ln = [int(x) for x in lens]
    n, total = len(ln), sum(ln)
    lo, hi, at, pre = [], [], 0, 0
    for r in range(world):
        lo.append(at)
        if r + 1 == world:
            hi.append(n)
            break
        target = total * (r + 1) // world
        while at < n and pre + ln[at] <= target:
            pre += ln[at]
            at += 1
        if at < n and pre < target and (target - pre) * 2 > ln[at]:
            pre += ln[at]
            at += 1
        hi.append(at)
    return lo, hi


def bitmap_words(n: int) -> int:
    return (n + 31) // 32


def pack_bitmap(accept: np.ndarray) -> np.ndarray:
    """accept bytes (0/1) -> little-endian bitmap of uint32 words (same layout as rsv_accept_bitmap_dev)."""
    n = len(accept)
    bits = np.zeros(bitmap_words(n) * 32, np.uint8)
    bits[:n] = accept != 0
    return np.packbits(bits, bitorder="little").view(np.uint32)


def unpack_bitmap(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n]


# ----------------------------------------------------------------------------------------------- exchange
class BitmapExchange:
    """The one collective step of a sharded batch.  Every rank contributes `slice_words` int32 words (the largest
    shard's bitmap size; smaller shards leave the tail zero) and a count; afterwards every rank holds all slices.

    Buffers are allocated once: `local` (this rank's slice — the verifier writes its bitmap straight into it),
    `gathered` (world x slice_words), `count` (1 x int64: local accepts in, job total out)."""

    def __init__(self, n_total: int, rank: int, world: int, dist, torch, device, plan=None):
        """plan = (lo, hi) of every rank (shard_plan's cuts, or the caller's): slices of unequal width, every rank sends the
        widest's words; None: the job is cut by shard_range."""
        self.n_total, self.rank, self.world, self.dist, self.torch = n_total, rank, world, dist, torch
        if plan is None:
            plan = tuple(zip(*[shard_range(n_total, r, world) for r in range(world)]))
        self.plan = ([int(x) for x in plan[0]], [int(x) for x in plan[1]])
        if len(self.plan[0]) != world or self.plan[0][0] != 0 or self.plan[1][-1] != n_total or \
                any(self.plan[0][r + 1] != self.plan[1][r] for r in range(world - 1)) or any(h < l for l, h in zip(*self.plan)):
            raise ValueError("a shard plan is contiguous from 0 to n_total, one [lo, hi) per rank")
        self.lo, self.hi = self.plan[0][rank], self.plan[1][rank]
        self.slice_words = max(1, max(bitmap_words(h - l) for l, h in zip(*self.plan)))  # the widest shard's
        self.device = device
        self.local = torch.zeros(self.slice_words, dtype=torch.int32, device=device)
        self.gathered = torch.zeros(world * self.slice_words, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        if world > 1 and not dist.is_initialized():
            raise RuntimeError("world > 1 needs an initialised process group (init_rank)")
        backend = dist.get_backend() if dist.is_initialized() else None
        # gloo moves host memory: stage device tensors through the host (CPU tests, one-GPU rehearsal)
        self.stage = backend == "gloo" and getattr(device, "type", str(device)) != "cpu"
        if self.stage:
            self.h_local = torch.zeros(self.slice_words, dtype=torch.int32)
            self.h_gathered = torch.zeros(world * self.slice_words, dtype=torch.int32)
            self.h_count = torch.zeros(1, dtype=torch.int64)
        self.collective = backend is not None

    def run(self):
        """all-gather `local` into `gathered`, all-reduce `count`.  The caller has ordered the producer of `local` /
        `count` before torch's current stream (Context.release_to_torch) — collectives are enqueued there."""
        if not self.collective:
            self.gathered.copy_(self.local)
            return
        if self.stage:
            self.h_local.copy_(self.local)
            self.h_count.copy_(self.count)
            self.dist.all_gather_into_tensor(self.h_gathered, self.h_local)
            self.dist.all_reduce(self.h_count)
            self.gathered.copy_(self.h_gathered)
            self.count.copy_(self.h_count)
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.local)
            self.dist.all_reduce(self.count)

    def assemble(self) -> np.ndarray:
        """The job's accept vector (uint8[n_total]) from the gathered slices; host side, outside the timed step."""
        g = self.gathered.cpu().numpy().view(np.uint32).reshape(self.world, self.slice_words)
        out = np.zeros(self.n_total, np.uint8)
        for r in range(self.world):
            lo, hi = self.plan[0][r], self.plan[1][r]
            out[lo:hi] = unpack_bitmap(g[r], hi - lo)
        return out

    def total_accepted(self) -> int:
        return int(self.count.item())


def gather_accept_bitmap(local_bitmap, n_total: int, rank: int, world: int, dist, torch, plan=None):
    """Functional form of BitmapExchange for a caller that already holds its shard's bitmap as an int32 tensor:
    returns the job's accept vector (uint8[n_total]) on every rank."""
    ex = BitmapExchange(n_total, rank, world, dist, torch, local_bitmap.device, plan=plan)
    ex.local[: local_bitmap.numel()] = local_bitmap
    ex.run()
    return ex.assemble()


class CExchange:
    """The same step through the C-ABI (rsv_exchange_*: ONE ncclAllGather + ONE ncclAllReduce issued by the library on the
    verifier's own stream, RCCL bound by dlopen) — what a Rust host would call; same buffers and `assemble()` as
    BitmapExchange.  The RCCL unique id travels through the torch process group (a host-side broadcast: any channel would
    do); torch.distributed carries nothing of the data path."""

    def __init__(self, rsv, ctx, n_total: int, rank: int, world: int, dist, torch, device, plan=None):
        self.n_total, self.rank, self.world, self.torch = n_total, rank, world, torch
        uid = [rsv.exchange_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        self.x = rsv.Exchange(ctx, uid[0], rank, world, n_total, plan=plan)
        if plan is None:
            plan = tuple(zip(*[shard_range(n_total, r, world) for r in range(world)]))
        self.plan = ([int(v) for v in plan[0]], [int(v) for v in plan[1]])
        self.lo, self.hi, self.slice_words = self.x.lo, self.x.hi, self.x.slice_words
        assert (self.lo, self.hi) == (self.plan[0][rank], self.plan[1][rank])
        self.local = torch.zeros(self.slice_words, dtype=torch.int32, device=device)
        self.gathered = torch.zeros(world * self.slice_words, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        self._rsv = rsv

    def run(self):
        self.x.run(self.local, self.gathered, self.count)  # enqueued on the context's stream, behind the verifying pass

    def assemble(self) -> np.ndarray:
        g = self.gathered.cpu().numpy().view(np.uint32).reshape(self.world, self.slice_words)
        return self._rsv.exchange_assemble(self.n_total, self.world, g, plan=self.plan)[0]

    def total_accepted(self) -> int:
        return int(self.count.item())

    def close(self):
        self.x.close()


class ShardedVerifier:
    """Rank `rank` of a `world`-rank verification job of `n_total` proofs on ONE GPU.

        sv = ShardedVerifier(rsv, n_total, rank, world, device_index, dist, torch)
        sv.step(d_blob, d_offsets, cfg)         # this rank's shard, resident in HBM
        accept = sv.exchange.assemble()           # whole job, every rank

    step() = rsv_verify_hints_dev on the shard (verdicts + the accept bitmap and count into the exchange slice) ->
    all-gather + all-reduce.  Nothing blocks the host: the verifier's stream is ordered before torch's current stream (on which
    the collectives are enqueued) with an event.
    plan: (lo, hi) of every rank — shard_plan's byte-balanced cuts for a job of mixed shapes; None: shard_range.
    exchange: "torch" (torch.distributed collectives; what the multi-rank tests rehearse) or "c" (rsv_exchange_* through the
    C-ABI, the library's own RCCL calls on the verifier's stream)."""

    def __init__(self, rsv, n_total: int, rank: int, world: int, device_index: int, dist, torch, plan=None, exchange: str = "torch"):
        self.rsv, self.torch = rsv, torch
        dev = torch.device("cuda", device_index)
        self.ctx = rsv.Context(device_index)
        self.c_exchange = exchange == "c"
        if self.c_exchange:
            self.exchange = CExchange(rsv, self.ctx, n_total, rank, world, dist, torch, dev, plan=plan)
        else:
            self.exchange = BitmapExchange(n_total, rank, world, dist, torch, dev, plan=plan)
        self.n_local = self.exchange.hi - self.exchange.lo
        self.d_accept = torch.zeros(max(self.n_local, 1), dtype=torch.uint8, device=dev)
        self.d_reason = torch.zeros(max(self.n_local, 1), dtype=torch.uint8, device=dev)

    def step(self, d_blob, d_offsets, cfg, inputs=None, hints=None):
        kw = {} if inputs is None else {"inputs": inputs}
        # verdicts, and the rank's bitmap slice + accept count straight into the exchange buffers, from ONE pass
        # (rsv_hints_out::d_accept_bitmap: the kernel that writes the verdicts packs them)
        self.ctx.verify_hints(d_blob, d_offsets, self.n_local, self.d_accept, self.d_reason, cfg=cfg, **kw, **(hints or {}),
                              d_accept_bitmap=self.exchange.local, d_accept_count=self.exchange.count)
        if not self.c_exchange:
            self.ctx.release_to_torch()
        self.exchange.run()

    def synchronize(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize()

    def close(self):
        if self.c_exchange:
            self.exchange.close()
        self.ctx.close()


# ----------------------------------------------------------------------------------------------- launcher
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(script: str, argv, n_ranks: int, env=None) -> int:
    """Start the `n_ranks` rank processes of a one-node job: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> script argv...` as a CHILD process, and return
    its exit code.  The calling process must not have touched the GPU (it only waits); it is never replaced."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=e)


def init_rank(torch, dist, rehearsal: bool = False, force_group: bool = False):
    """Per-rank setup from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, the launcher's
    contract): (rank, world, device_index).  Backend "nccl" (= RCCL) with the rank's own GPU; in rehearsal mode (a
    one-GPU box) every rank uses cuda:0 and the exchange runs over gloo, since RCCL refuses two ranks on one device.
    force_group: a process group even at world size 1 (the tests drive the real RCCL backend that way)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1 or force_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    return rank, world, dev_index
