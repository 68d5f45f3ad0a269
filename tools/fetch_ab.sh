#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the tree kernels under the two workgroup orders
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for K in 2 1; do for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$K -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --perm-log2 0 --witness-proofs 0 --no-single-proof --knob tree_order=$K > $OUT/b_${C}_$K.json 2> $OUT/b_${C}_$K.err
done; done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for K in (2, 1):
    tot = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        p = glob.glob(f"{out}/pmc_{C}_{K}/**/*counter_collection.csv", recursive=True)[0]
        per = {}
        for r in csv.DictReader(open(p)):
            if "rsv::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rsv::", "")
            per.setdefault((k, int(r["Dispatch_Id"])), 0.0)
            per[(k, int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
        last = {}
        for (k, d), v in per.items(): last[k] = (d, v) if k not in last or d > last[k][0] else last[k]
        for k, (d, v) in last.items(): tot.setdefault(k, {})[C] = v
    s = 0
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        b = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
        s += b
        if b > 5e7: print(f"tree_order={K} {k:40s} corrected {b / 1e9:6.3f} GB")
    print(f"tree_order={K} total {s / 1e9:6.3f} GB")
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
