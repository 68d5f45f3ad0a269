#!/bin/bash
# HBM traffic of the witness kernels (PMC, two separate passes; the program itself follows `--`):
#   gpurun --timeout 600 -- 'bash tools/witness_pmc.sh'   ->  gpurun_out/witness_pmc/summary.txt
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/witness_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 280 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$ROOT/tools/bench_witness.py" --fixture level10-1.bin --proofs 4096 --steps 1 --warmup 0 > "$OUT/bench_$C.json" 2> "$OUT/pmc_$C.err"
    echo "pmc $C done"
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json, collections
out = "gpurun_out/witness_pmc"
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    tot[c] = (acc, calls)
d = json.load(open(f"{out}/bench_FETCH_SIZE.json"))
n, nv = d["config"]["proofs"], d["config"]["variables_per_proof"]
lines = [f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) of tools/bench_witness.py --fixture level10-1.bin --proofs {n} --steps 1 --warmup 0",
         "# the tool makes 2 witness calls (the timed one + the verdict check is the same call; then 1 hints-only call): sums over ALL dispatches of a kernel",
         "# hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream, MI355X_MICROARCH.md)",
         "kernel,dispatches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_corrected"]
for k in sorted(tot["FETCH_SIZE"][0]):
    if "witness" not in k: continue
    fz, wz = tot["FETCH_SIZE"][0][k], tot["WRITE_SIZE"][0].get(k, 0.0)
    lines.append(f"{k},{len(tot['FETCH_SIZE'][1][k])},{fz:.0f},{wz:.0f},{(2 * fz + wz) * 1024:.0f}")
lines.append(f"# algorithmic bytes per witness call: {d['roofline']['note']}; x {n} proofs")
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
