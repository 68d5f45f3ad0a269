"""N>1 host path on CPU: gloo ranks shard a job, verify their shards and run the product's exchange
(recursive-stwo_amd/sharding.py: BitmapExchange — the object ShardedVerifier.step() and bench.py use).
The verifier used inside the CPU ranks is the oracle (test infrastructure); the GPU product path through the very
same exchange is covered by tests/test_multi_gpu.py (-m gpu).  This test covers the partition, the bitmap layout,
the all-gather of unequal shards, the all-reduce of the count and the launcher's command line."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_sharding():
    spec = importlib.util.spec_from_file_location("rsv_sharding", os.path.join(ROOT, "recursive-stwo_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    return sh


def _plan_for(sh, n_total, world, planned):
    """planned: a byte-balanced plan over synthetic lengths that fall off steeply (the level-ordered chain in small), so that
    the shards are of very unequal COUNT: the widest slice sets slice_words, the narrow ones leave most of theirs zero."""
    if not planned:
        return None
    lens = [4000 // (1 + i // 3) + 7 for i in range(n_total)]
    return sh.shard_plan(lens, world)


def _worker(rank, world, port, n_total, q, planned=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    sh = load_sharding()
    from tests import oracle_binding as ob
    from tests.conftest import read_proof

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    proof = read_proof("small_proof.bin")
    plan = _plan_for(sh, n_total, world, planned)
    ex = sh.BitmapExchange(n_total, rank, world, dist, torch, torch.device("cpu"), plan=plan)
    batch = [ob.tamper(proof, i) if i % 5 == 2 else proof for i in range(ex.lo, ex.hi)]
    acc, _ = ob.verify_batch(batch, ob.PcsConfig(20, 5, 2, 16), [(1, (1, 0, 0, 0))])
    bits = sh.pack_bitmap(acc).view(np.int32)
    results = []
    for _ in range(2):  # the buffers are reused step after step
        ex.local.zero_()
        ex.local[: bits.size] = torch.from_numpy(bits.copy())
        ex.count[0] = int(acc.sum())
        ex.run()
        results.append((ex.assemble().tolist(), ex.total_accepted()))
    # the functional form on the same group
    full = sh.gather_accept_bitmap(torch.from_numpy(bits.copy()), n_total, rank, world, dist, torch, plan=plan)
    q.put((rank, results, full.tolist()))
    dist.destroy_process_group()


def test_shard_range_partitions():
    sh = load_sharding()
    for n in (0, 1, 7, 8, 65536, 1048576 + 3):
        for w in (1, 2, 3, 8):
            spans = [sh.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    acc = np.array([1, 0, 1, 1, 0] * 13, np.uint8)
    assert sh.unpack_bitmap(sh.pack_bitmap(acc), len(acc)).tolist() == acc.tolist()
    with pytest.raises(ValueError):
        sh.shard_range(4, 2, 2)


def test_single_rank_exchange_without_process_group():
    """world == 1 and no process group: run() is a local copy (what `python bench.py` does at N = 1)."""
    import torch
    import torch.distributed as dist
    sh = load_sharding()
    ex = sh.BitmapExchange(70, 0, 1, dist, torch, torch.device("cpu"))
    acc = (np.arange(70) % 3 != 0).astype(np.uint8)
    ex.local[:] = torch.from_numpy(sh.pack_bitmap(acc).view(np.int32).copy())
    ex.count[0] = int(acc.sum())
    ex.run()
    assert ex.assemble().tolist() == acc.tolist() and ex.total_accepted() == int(acc.sum())
    with pytest.raises(RuntimeError):
        sh.BitmapExchange(70, 0, 2, dist, torch, torch.device("cpu"))  # N > 1 needs init_rank first


def test_launcher_command_line(monkeypatch):
    """`python bench.py --gpus N` as a plain process: the parent starts `python -m torch.distributed.run` with the
    driver's flags as a CHILD (subprocess, never exec) and returns its exit code."""
    sh = load_sharding()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(sh.subprocess, "call", fake_call)
    assert sh.launch_ranks("/x/bench.py", ["--gpus", "4", "--steps", "2"], 4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_shard_plan_restatement_and_balance():
    """sharding.shard_plan (Python) == rsv_shard_plan (the C entry point, host arithmetic: no device needed) on random and
    degenerate jobs; and the property it exists for: on the reference's chain in LEVEL order (13 x 4 096 proofs, 435 KB
    down to 76 KB) over 8 ranks the heaviest shard carries <= 1.1 x the lightest's bytes (by count: 4.5 x)."""
    sys.path.insert(0, ROOT)
    import rsvload
    rsv = rsvload.load_package()
    sh = load_sharding()
    rng = np.random.default_rng(7)
    for trial in range(60):
        n = int(rng.integers(0, 400))
        world = int(rng.integers(1, 12))
        lens = rng.integers(0, 500000, n) if trial % 3 else np.repeat(rng.integers(1, 500000, 5), 80)[:n]
        assert sh.shard_plan(lens, world) == rsv.shard_plan(lens.astype(np.uint64), world), (trial, n, world)
    from tests.conftest import read_proof
    names = ["level1-5.bin", "level2-1.bin", "level3-1.bin", "level4-5.bin", "level5-1.bin", "level6-1.bin", "level7-1.bin",
             "level8-1.bin", "level9-1.bin", "level10-1.bin", "level11-1.bin", "level12-1.bin", "level13-1.bin"]
    lens = np.repeat([len(read_proof(nm)) for nm in names], 4096)
    lo, hi = sh.shard_plan(lens, 8)
    pre = np.concatenate([[0], np.cumsum(lens)])
    b = [int(pre[h] - pre[l]) for l, h in zip(lo, hi)]
    assert max(b) / min(b) <= 1.1 and max(h - l for l, h in zip(lo, hi)) > 2 * min(h - l for l, h in zip(lo, hi))
    with pytest.raises(ValueError):
        import torch
        import torch.distributed as dist
        sh.BitmapExchange(70, 0, 1, dist, torch, torch.device("cpu"), plan=([0], [69]))  # a plan covers the whole job


@pytest.mark.timeout(240)
@pytest.mark.parametrize("n_total,world,planned", [(23, 2, False), (64, 2, False), (7, 3, False), (83, 8, False), (5, 8, False),
                                                   (83, 8, True), (65, 2, True), (40, 3, True)])
def test_gloo_ranks_bitmap_exchange(n_total, world, planned):
    """planned: the job cut by shard_plan over lengths that fall off steeply — slices of very unequal width (83 proofs over 8
    ranks: 3 ... 40 proofs), every rank sending the widest slice's words."""
    import torch.multiprocessing as mp
    sh = load_sharding()
    if planned:
        lo, hi = _plan_for(sh, n_total, world, True)
        assert max(h - l for l, h in zip(lo, hi)) >= 3 * max(1, min(h - l for l, h in zip(lo, hi)))
    ctx = mp.get_context("spawn")
    port = sh.free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q, planned)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = [0 if i % 5 == 2 else 1 for i in range(n_total)]
    for rank, steps, full in results:
        for acc, count in steps:
            assert acc == want, rank
            assert count == sum(want)
        assert full == want
