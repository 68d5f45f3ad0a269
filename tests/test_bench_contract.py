"""bench.py contract: one JSON line with the fields the driver reads; the N>1 path is rehearsed with two
ranks on one GPU (`--rehearsal`: bitmap exchange over gloo, since RCCL refuses two ranks per device), here in
the explicit torchrun form the driver uses for N > 1 (tests/test_multi_gpu.py covers the self-launching form)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_line(out, n_gpus):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    # what binds the dominant kernel is integer VALU issue; achieved / peak / frac stay the tier's nominal HBM figure
    assert r["bound"] == "valu" and r["nominal_bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["valu_issue_frac"] is None or 0 < r["valu_issue_frac"] <= 1.0
    assert r["pipeline_valu_issue_frac"] is None or 0 < r["pipeline_valu_issue_frac"] <= 1.0
    assert 0 < r["pipeline_frac"] <= r["frac"] and (r["traffic"] is None or r["traffic"] > 0)
    assert d["value"] > 0 and d["ms_per_step"] > 0
    return d


def test_bench_refuses_without_gpu():
    import rsvload
    if rsvload.load_package().device_count() > 0:
        pytest.skip("a HIP device is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--proofs", "64"], capture_output=True, text=True)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--proofs", "2048", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "64", "--perm-log2", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check_line(out.stdout, 1)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert 4000 < d["cpu_baseline"]["perms_per_proof"] < 6000
    assert d["cpu_baseline"]["one_thread"]["value"] > 0 and d["cpu_baseline"]["one_thread"]["cores"] == 1 and d["cpu_baseline"]["cpu_model"]
    # the headline CPU figure uses every CPU the process may use (all host cores; min of logical CPUs,
    # affinity and cgroup quota), and says what it found
    share = d["cpu_baseline"]["cpu_share"]
    assert share["usable_cpus"] >= 1 and d["cpu_baseline"]["cores"] == min(share["usable_cpus"], 64)
    assert d["host_path"]["value"] > 0 and d["host_path"]["GBps"] > 0  # PCIe-inclusive rate, reported beside `value`
    assert d["host_path"]["pinned_arena"]["GBps"] > 0                   # the same from an rsv_host_alloc arena (no gather copy)
    assert 0 < d["single_proof"]["min_ms"] <= d["single_proof"]["latency_ms"] < 50   # one proof per call: the reference's own use
    assert d["config"]["exchange"]["world_size"] == 1 and len(d["config"]["exchange"]["devices"]) == 1
    v = d["valu"]
    assert 0 < v["frac_of_ceiling"] <= 1.0 and 0 < v["pipeline_frac_of_ceiling"] <= 1.0
    assert v["ceiling_perms_per_s"] > v["poseidon2_perms_per_s"] > 1e9
    e = d["emulated_poseidon2"]
    assert e["bound"] == "hbm" and 0 < e["frac"] <= 1.0 and e["algorithmic_bytes_per_perm"] == 65 + 416 * 16
    w = d["recursion_circuit_witness"]
    assert w["proofs"] == 1024 and w["variables_per_proof"] == 47788 and w["levels"] > 100 and w["proofs_per_s"] > 1e4
    assert w["cpu_baseline"]["kind"] == "port" and 0 < w["cpu_baseline"]["value"] < w["proofs_per_s"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--proofs", "2048",
           "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--perm-log2", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _check_line(out.stdout, 2)
    assert d["config"]["parallelism"] == "shard2"


def test_committed_profile_holds_the_large_batch_line():
    """VERDICT r3 #6 — hold, don't dig: the committed profile of the default workload (profiles/pmc_latest.json, written by
    tools/profile.sh + tools/pmc_summary.py from a gpurun of `python bench.py` on one MI355X) must show >= 1.90 M proofs/s
    and <= 2.6 x the algorithmic bytes in HBM traffic over the pipeline; a change that costs the large batch more than
    that does not get committed with a fresh profile."""
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
        prof = json.load(f)
    if "pipeline_traffic_ratio" not in prof:
        pytest.skip("profile written before round 4")
    assert prof["proofs"] == 65536 and prof["bench_value_proofs_per_s"] >= 1.90e6, prof["bench_value_proofs_per_s"]
    assert prof["pipeline_traffic_ratio"] <= 2.6, prof["pipeline_traffic_ratio"]
    dom = prof["kernels"]["k_pair_merkle"]
    # the FRI trees' span: 21-22 ms while k_query ended ~8 ms into the step; ~25 since it ends at ~4.5 ms and the trees start
    # beside the trace trees that much earlier (same step time: the whole-step figure below is the one that must hold)
    assert dom["avg_ms"] < 27.0 and dom["SQ_INSTS_VALU"] > 1e10
    if "pipeline_valu_insts" in prof:
        frac = prof["pipeline_valu_insts"] / (prof["bench_ms_per_step"] * 1e-3) / (1024 * 2.4e9 / 2.0)
        assert 0.5 <= frac <= 1.0, frac
    # the evidence is ONE build — the profile and the large soak carry the hash of the same kernel sources
    # (tests/soak.py prints it; tools/round5_artifacts.sh runs the soak last), and those are the committed sources
    tag = prof["tag"].split("_")[0]
    soak = os.path.join(ROOT, "profiles", f"{tag}_soak_large.txt")
    if tag >= "r5":
        import re
        shas = set(re.findall(r"kernel sources ([0-9a-f]{16})", open(soak).read()))
        assert shas == {prof["kernel_sources_sha"]}, (shas, prof["kernel_sources_sha"])
        sys.path.insert(0, ROOT)
        import bench
        assert bench.kernel_sources_sha() == prof["kernel_sources_sha"], "profiles/ were taken on other kernel sources than the committed ones"
