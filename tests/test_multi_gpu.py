"""The N-GPU path on the one GPU of the test box (-m gpu):
  * the REAL RCCL backend ("nccl") at world_size 1: init_process_group with device_id, ShardedVerifier.step()
    (verify -> bitmap -> all_gather_into_tensor + all_reduce on device tensors), barrier, destroy;
  * `python3 bench.py --gpus 2` started exactly as the driver starts the N = 1 run — a plain process that must
    launch its own ranks — in rehearsal mode (both ranks on cuda:0, exchange over gloo: RCCL refuses two ranks on
    one device);
  * `--total-proofs` (strong scaling, BASELINE configs[3] shape) through the same launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NCCL_RANK = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["RSV_ROOT"])
import rsvload
rsv = rsvload.load_package()
from recursive_stwo_amd import sharding
import torch, torch.distributed as dist
from tests import oracle_binding as ob
from tests.conftest import read_proof, fixture_cfg

os.environ["RSV_FORCE_PROCESS_GROUP"] = "1"          # world_size 1 still goes through RCCL
os.environ.setdefault("MASTER_PORT", str(sharding.free_port()))
rank, world, dev_index = sharding.init_rank(torch, dist)
assert dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", dev_index)
proof = read_proof("recursive_proof_16_15.bin")
n = 600
batch = [ob.tamper(proof, i) if i % 9 == 4 else proof for i in range(n)]
blob, offsets = rsv.pack(batch)
d_blob = torch.from_numpy(blob.copy()).to(dev)
d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
sv = sharding.ShardedVerifier(rsv, n, rank, world, dev_index, dist, torch)
assert not sv.exchange.stage and sv.exchange.collective
cfg = fixture_cfg("recursive_proof_16_15.bin")
for _ in range(3):
    sv.step(d_blob, d_off, cfg)
sv.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
want = np.array([0 if i % 9 == 4 else 1 for i in range(n)], np.uint8)
got = sv.exchange.assemble()
print(json.dumps({"nccl": "ok", "match": bool(np.array_equal(got, want)), "count": sv.exchange.total_accepted(),
                  "want_count": int(want.sum()), "gathered_on": str(sv.exchange.gathered.device), "t": float(t.item())}))
dist.destroy_process_group()
'''


def test_nccl_backend_world_size_one(tmp_path):
    script = tmp_path / "nccl_rank.py"
    script.write_text(NCCL_RANK)
    env = dict(os.environ, RSV_ROOT=ROOT, MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["nccl"] == "ok" and d["match"] and d["count"] == d["want_count"] and d["gathered_on"].startswith("cuda")


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """The driver's shape: `python3 bench.py --gpus 2 --steps K --warmup W` as ONE plain process."""
    d = _bench(["--gpus", "2", "--proofs", "2048", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--perm-log2", "0"],
               {"RSV_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "shard2"
    ex = d["config"]["exchange"]
    assert d["config"]["proofs_per_step"] == 4096 and d["value"] > 0 and "gloo" in ex["collectives"]
    # the line proves from inside the group that two ranks took part (here: two processes on the one device)
    assert ex["world_size"] == 2 and ex["backend"] == "gloo" and sorted(x["rank"] for x in ex["devices"]) == [0, 1]
    assert len({x["pid"] for x in ex["devices"]}) == 2


def test_bench_total_proofs_strong_scaling():
    d = _bench(["--gpus", "2", "--total-proofs", "4099", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--perm-log2", "0"],
               {"RSV_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["proofs_per_step"] == 4099
    assert d["config"]["proofs_rank0"] == 2050
    d1 = _bench(["--total-proofs", "3000", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--perm-log2", "0"], {})
    assert d1["n_gpus"] == 1 and d1["scaling"] == "strong" and d1["config"]["proofs_rank0"] == 3000
