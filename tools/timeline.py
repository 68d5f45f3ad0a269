#!/usr/bin/env python3
"""Timeline of the LAST verify step in a rocprofv3 --kernel-trace CSV: kernel, stream/queue, start and end in us relative
to the step's first kernel (k_parse or the front half of a split transcript).  Usage: python tools/timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rsv::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_parse starts the last step (a split transcript's front half may start a little earlier)
last_parse = max(i for i, r in enumerate(rows) if "k_parse" in r["Kernel_Name"])
first = last_parse
while first > 0 and ("k_transcript_row<1" in rows[first - 1]["Kernel_Name"] or "k_transcript<false, 1" in rows[first - 1]["Kernel_Name"]):
    first -= 1
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
run = None  # consecutive launches of one kernel (the witness program's levels) are printed as one line


def flush():
    if run:
        name, q, s, e, busy, count = run
        extra = f"  x{count} launches, {busy:.1f} us busy" if count > 1 else ""
        print(f"{name:28s} q{q:>3s} {s:9.1f} -> {e:9.1f}  ({e - s:8.1f} us){extra}")


for r in rows[first:]:
    if "k_permute" in r["Kernel_Name"] or "k_emulated" in r["Kernel_Name"]:
        break  # (the microbenchmarks bench.py runs behind the timed steps)
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rsv::", "").replace("(anonymous namespace)::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "?")
    if run and run[0] == name and run[1] == q:
        run = (name, q, run[2], e, run[4] + (e - s), run[5] + 1)
    else:
        flush()
        run = (name, q, s, e, e - s, 1)
flush()
