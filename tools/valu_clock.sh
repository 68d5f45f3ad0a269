#!/bin/bash
# VERDICT r4 #5b: is the fast class's "2.5 cycles" 2.0 shader cycles at a 1.9 GHz clock, or 2.5 cycles at 2.4 GHz?
#   gpurun -- 'bash tools/valu_clock.sh r5_lab'
# tools/valu_lab prints, per instruction, wall-normalised cycles (launch time x 2.4 GHz), shader cycles (clock64 around every
# wave's loop) and the clock the two imply, at 1 / 4 / 8 waves per SIMD; a rocprofv3 --pmc GRBM_GUI_ACTIVE pass of the
# 4-wave run gives the clock from the outside (GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration).  The program itself
# follows `--` (no shell hop under the profiler).
set -o pipefail
TAG=${1:-r5_lab}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for W in 1 4 8; do
    timeout -k 10 200 "$ROOT/tools/valu_lab" $W > "$OUT/valu_lab_${W}waves.txt" 2> "$OUT/valu_lab_${W}waves.err" && echo "lab $W waves done"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lab" -o pmc -- "$ROOT/tools/valu_lab" 4 > "$OUT/valu_lab_4waves_under_pmc.txt" 2> "$OUT/pmc_lab.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
path = glob.glob(out + "/pmc_lab/**/*counter_collection.csv", recursive=True)
if not path:
    print("no counter file"); sys.exit(0)
per = collections.OrderedDict()
for r in csv.DictReader(open(path[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    k = (r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"]))
    d = per.setdefault(k, [0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    d[0] += float(r["Counter_Value"])
last = collections.OrderedDict()
for (name, disp), (gui, ns) in per.items():
    last[name] = (gui, ns)
with open(out + "/valu_lab_grbm_clock.txt", "w") as f:
    f.write("# rocprofv3 --pmc GRBM_GUI_ACTIVE -- tools/valu_lab 4: clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (last dispatch of each kernel)\n")
    for name, (gui, ns) in last.items():
        f.write(f"{name:28s} {gui / 8 / ns:6.3f} GHz  ({ns / 1e6:7.3f} ms)\n")
print(open(out + "/valu_lab_grbm_clock.txt").read()[:1500])
PY
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
