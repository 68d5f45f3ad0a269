"""One-off soak: GPU verdict / reason against the oracle over a large mutant corpus of every Poseidon-channel fixture
(tests/mutants.py generators, several hundred random corruptions each), in mixed batches.  python tests/soak.py [n_random] [seed] [pow0|-] [single K]

pow0: every fixture's pow_bits header word is set to 0 and it is verified under pow_bits = 0, so that mutants of the
transcript-absorbed sections (commitments, sampled values, FRI layer commitments, last-layer polynomial) are not all
stopped by the proof of work: they reach the logup / composition checks, and — with query positions that no longer
match the decommitments — the plan, Merkle and FRI kernels.
single K: additionally verify the first K proofs of the shuffled corpus ONE PER CALL (uniform-batch path, natural slot
order, the small-batch kernel forms), in groups of 3 under the smallest workspace budget the library accepts (1 MB), and
the first 8 K proofs in one-configuration calls of 5..64 proofs (device-side slot order, row form of the tree kernels).  The
kernel forms can be forced for every call with trailing name=value arguments (names of rsv.OPTIONS, e.g.
transcript_form=lane oods_form=row plan_form=serial): they become process defaults (rsv_ctx_set_option, ctx = NULL)."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsvload  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests.mutants import mutants_of  # noqa: E402

rsv = rsvload.load_package()


def main():
    import bench  # (the repo root is on the path: the kernel sources' hash names the build this corpus ran on)
    print(f"# tests/soak.py {' '.join(sys.argv[1:])} — kernel sources {bench.kernel_sources_sha()}", flush=True)
    for a in [a for a in sys.argv[1:] if "=" in a]:  # kernel-form overrides
        name, value = a.split("=", 1)
        rsv.set_default_option(name, value if value in rsv.OPTION_VALUES else int(value))
        sys.argv.remove(a)
    n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["proofs"]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    rng = np.random.default_rng(seed)
    pow0 = len(sys.argv) > 3 and sys.argv[3] == "pow0"
    batch, cfgs = [], []
    for e in man:
        if [(i, tuple(v)) for i, v in e["inputs"]] != list(ob.STANDARD_INPUTS) or "struct" in e:
            continue  # (bitcoin_proof.bin is another proof struct: the mutant generators cannot re-serialize it)
        proof = open(os.path.join(ROOT, "tests", "golden", "proofs", e["file"]), "rb").read()
        if pow0:
            w = np.frombuffer(proof, np.uint32).copy()
            w[10] = 0  # W_POW_BITS
            proof = w.tobytes()
            e = dict(e, pow_bits=0)
        mut = mutants_of(proof, rng, n_random) + [proof]
        batch += mut
        # the reference's configuration literal of the fixture each mutant derives from (never the mutant's own header)
        cfgs += [ob.PcsConfig(e["pow_bits"], e["log_blowup_factor"], e["log_last_layer_degree_bound"], e["n_queries"])] * len(mut)
    order = rng.permutation(len(batch))
    batch = [batch[i] for i in order]
    cfgs = [cfgs[i] for i in order]
    print(f"{len(batch)} mutants built", flush=True)  # (a run that prints nothing for minutes is taken to be hung)
    t0 = time.perf_counter()
    acc, reason = rsv.verify_batch(batch, cfgs)
    t1 = time.perf_counter()
    print(f"GPU done in {t1 - t0:.2f} s", flush=True)
    parts = np.array_split(np.arange(len(batch)), 64)

    def judge(ix):
        r = ob.verify_batch([batch[i] for i in ix], [cfgs[i] for i in ix])
        print(".", end="", flush=True)
        return r

    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(judge, parts))
    print(flush=True)
    oacc = np.concatenate([r[0] for r in res])
    oreason = np.concatenate([r[1] for r in res])
    t2 = time.perf_counter()
    diff = np.nonzero((acc != oacc) | (reason != oreason))[0]
    print(f"{len(batch)} proofs: GPU {t1 - t0:.2f} s (host path), oracle {t2 - t1:.1f} s; accepted {int(acc.sum())}; "
          f"reasons {np.bincount(reason, minlength=13).tolist()}; mismatches {diff.size}")
    for i in diff[:10]:
        print("  mismatch", int(i), int(acc[i]), int(reason[i]), int(oacc[i]), int(oreason[i]), len(batch[i]))
    bad = int(diff.size)
    if len(sys.argv) > 5 and sys.argv[4] == "single":
        k = min(int(sys.argv[5]), len(batch))
        one = 0
        for i in range(k):
            a, r = rsv.verify_batch([batch[i]], [cfgs[i]])
            one += int(a[0] != oacc[i] or r[0] != oreason[i])
        rsv.set_default_option("ws_budget_mb", 1)
        three = 0
        for i in range(0, k - 2, 3):
            a, r = rsv.verify_batch(batch[i:i + 3], cfgs[i:i + 3])
            three += int((a != oacc[i:i + 3]).any() or (r != oreason[i:i + 3]).any())
        rsv.set_default_option("ws_budget_mb", 8192)
        print(f"single: {k} one-proof calls, mismatches {one}; {k // 3} three-proof calls under the minimum workspace budget, mismatches {three}")
        bad += one + three
        # one configuration per call, 5..64 proofs: the device-side slot order and (up to 32 queries) the row form of the
        # tree kernels, on the first 8 k proofs of the corpus
        by_cfg = {}
        for i in range(min(8 * k, len(batch))):
            c = cfgs[i]
            by_cfg.setdefault((c.pow_bits, c.log_blowup_factor, c.log_last_layer_degree_bound, c.n_queries), []).append(i)
        small = calls = 0
        for ix in by_cfg.values():
            at = 0
            while at < len(ix):
                take = ix[at:at + int(rng.integers(5, 65))]
                at += len(take)
                a, r = rsv.verify_batch([batch[i] for i in take], cfgs[take[0]])
                small += int((a != oacc[take]).any() or (r != oreason[take]).any())
                calls += 1
        print(f"small: {calls} one-configuration calls of 5..64 proofs, mismatches {small}")
        bad += small
        # every configuration's proofs in ONE call: with trailing tree_pace=row16 query_form=row this is the row forms on
        # launches of hundreds to thousands of proofs (many workgroups, the in-kernel cap, k_cap_top behind it)
        whole = 0
        for ix in by_cfg.values():
            a, r = rsv.verify_batch([batch[i] for i in ix], cfgs[ix[0]])
            whole += int((a != oacc[ix]).sum() + ((a == oacc[ix]) & (r != oreason[ix])).sum())
        print(f"whole: {len(by_cfg)} one-configuration calls of {min(map(len, by_cfg.values()))}..{max(map(len, by_cfg.values()))} proofs, mismatching proofs {whole}")
        bad += whole
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
