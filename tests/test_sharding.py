"""N>1 host path on CPU: two gloo ranks shard a batch, verify their shards and all-gather the accept bitmaps.
The verifier used inside the ranks is the CPU oracle (test infrastructure) — the GPU product path is
exercised by tests/test_gpu_parity.py; this test covers sharding, bitmap packing and the collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import importlib.util
    spec = importlib.util.spec_from_file_location("rsv_sharding", os.path.join(ROOT, "recursive-stwo_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    from tests import oracle_binding as ob
    from tests.conftest import read_proof

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    proof = read_proof("small_proof.bin")
    lo, hi = sh.shard_range(n_total, rank, world)
    batch = [ob.tamper(proof, i) if i % 5 == 2 else proof for i in range(lo, hi)]
    acc, _ = ob.verify_batch(batch, [(1, (1, 0, 0, 0))])
    local = torch.from_numpy(sh.pack_bitmap(acc).view(np.int32).copy())
    full = sh.gather_accept_bitmap(local, n_total, rank, world, dist, torch)
    count = torch.tensor([int(acc.sum())], dtype=torch.int64)
    dist.all_reduce(count)
    q.put((rank, full.tolist(), int(count.item())))
    dist.destroy_process_group()


def test_shard_range_partitions():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rsv_sharding", os.path.join(ROOT, "recursive-stwo_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
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


@pytest.mark.timeout(180)
def test_two_rank_gloo_bitmap_exchange():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    n_total, world = 23, 2
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = [0 if i % 5 == 2 else 1 for i in range(n_total)]
    for rank, full, count in results:
        assert full == want, rank
        assert count == sum(want)
