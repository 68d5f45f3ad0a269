"""Full-size parity (`-m gpu`): the DEFAULT code path — no tuning knob touched — at BASELINE.json's sizes, accept AND
reason of every proof against the oracle's verdict on the distinct inputs.

The batches are built by bench.py's own builder (round-robin over the reference's fixtures, proof i with i % 17 == 5
gets one flipped bit), so what is checked here is exactly what the bench line times:
  (a) BASELINE configs[2]: 65 536 proofs of the multi-proofs standard configuration (4 fixtures, 3 tree geometries)
  (b) BASELINE configs[4]: the recursion chain, 13 shapes / 6 PCS configurations, 4 096 proofs each in ONE mixed batch
  (c) BASELINE configs[3]: one GPU's shard of the 1 M-proof job, 131 072 proofs = 15.4 GB (blob offsets far beyond
      4 GB, grids beyond 65 535 workgroups), and the same shard with the per-query workspace cut into groups.
Everything above 24 576 proofs — the lane forms of the transcript / OODS / quotient-constant kernels, launch grids in
the hundreds of thousands of workgroups — is reached here without an override; the small-batch forms are covered by
tests/test_gpu_parity.py.  The oracle (16 threads) judges only the distinct inputs: the genuine fixtures and the
tampered copies, which are read back from HBM so that it sees the very bytes the GPU verified.
Reference: examples/multi-proofs/src/main.rs:173-295 (configurations per level)."""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import bench
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _oracle(proofs, cfgs, threads=16):
    parts = [ix for ix in np.array_split(np.arange(len(proofs)), threads) if len(ix)]
    with ThreadPoolExecutor(len(parts)) as ex:
        res = list(ex.map(lambda ix: ob.verify_batch([proofs[i] for i in ix], [cfgs[i] for i in ix]), parts))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def _run(rsv, n, first_index, fixtures, budgets=(None,)):
    import torch
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    d_blob, d_offsets, plen, tam, fix_idx = bench.build_batch_on_device(torch, dev, n, first_index, fixtures)
    fcfg = bench.fixture_configs(rsv, fixtures)
    # the oracle's verdict on the distinct inputs: every genuine fixture, every tampered copy (bytes read back from HBM)
    offs = d_offsets.cpu().numpy()
    tampered = [bytes(d_blob[int(offs[i]):int(offs[i + 1])].cpu().numpy()) for i in tam]
    genuine = [bench.read_fixture(f) for f in fixtures]
    for i in tam[:: max(1, len(tam) // 50)]:  # the tamper rule really changed them
        assert bytes(d_blob[int(offs[i]):int(offs[i + 1])].cpu().numpy()) != genuine[fix_idx[i]]
    oacc, oreason = _oracle(genuine + tampered, fcfg + [fcfg[fix_idx[i]] for i in tam])
    assert oacc[:len(genuine)].all(), "the oracle accepts every reference fixture"
    want_acc = oacc[:len(genuine)][fix_idx].astype(np.uint8)
    want_reason = oreason[:len(genuine)][fix_idx].astype(np.uint8)
    want_acc[tam], want_reason[tam] = oacc[len(genuine):], oreason[len(genuine):]
    assert int(want_acc.sum()) == n - len(tam), "every single-bit tamper is rejected by the oracle"
    t1 = time.perf_counter()
    for budget in budgets:
        ctx = rsv.Context(0)
        if budget is not None:  # the only knob any of these tests touches, and only in the second pass of (c)
            ctx.set_option("ws_budget_mb", budget)
        multi = len({rsv._cfg_key(c) for c in fcfg}) > 1
        cfg = ctx.prepare_cfg([fcfg[k] for k in fix_idx], n) if multi else rsv.PreparedCfg([fcfg[0]])
        d_acc = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        d_reason = torch.full((n,), 99, dtype=torch.uint8, device=dev)
        ctx.verify_batch(d_blob, d_offsets, n, d_acc, d_reason, cfg=cfg)
        ctx.synchronize()
        acc, reason = d_acc.cpu().numpy(), d_reason.cpu().numpy()
        bad = np.nonzero((acc != want_acc) | (reason != want_reason))[0]
        assert bad.size == 0, (budget, bad[:10].tolist(), acc[bad[:10]].tolist(), reason[bad[:10]].tolist(),
                               want_acc[bad[:10]].tolist(), want_reason[bad[:10]].tolist())
        ctx.close()
    t2 = time.perf_counter()
    reasons = np.bincount(want_reason, minlength=13).tolist()
    print(f"{n} proofs, {int(offs[-1]) / 1e9:.2f} GB, {len(tam)} tampered: build + oracle {t1 - t0:.1f} s, GPU {t2 - t1:.1f} s, reasons {reasons}")
    del d_blob
    torch.cuda.empty_cache()
    return reasons


def test_configs2_standard_mix_65536(rsv):
    """BASELINE configs[2], the bench line's own batch: 65 536 proofs, 7.7 GB."""
    reasons = _run(rsv, 65536, 0, bench.WORKLOADS["standard"])
    assert reasons[0] == 65536 - len(range(5, 65536, 17)) and sum(reasons[6:12]) > 1000  # most flips land in Merkle witnesses


def test_configs4_recursion_chain_53248_mixed(rsv):
    """BASELINE configs[4]: the 13 chain fixtures x 4 096 in one mixed batch (6 configurations, n_queries 8..80)."""
    reasons = _run(rsv, 13 * 4096, 0, bench.WORKLOADS["chain"])
    assert reasons[0] == 13 * 4096 - len(range(5, 13 * 4096, 17))


def test_configs3_shard_131072(rsv):
    """BASELINE configs[3]: rank 3's contiguous shard of the 1 048 576-proof job (global proof indices 393 216 ..),
    15.4 GB resident; then the same shard with the per-query workspace held to 512 MB (cut into groups)."""
    _run(rsv, 131072, 3 * 131072, bench.WORKLOADS["standard"], budgets=(None, 512))
