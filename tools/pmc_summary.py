"""gpurun_out/<tag>/ (tools/profile.sh) -> profiles/<tag>_{bench_n1_65536.json, bench_under_rocprof.json,
kernel_stats_bench65536.csv, pmc_summary.csv} and profiles/pmc_latest.json (what bench.py reads for roofline.traffic:
keyed by a hash of the kernel sources that were profiled, so a stale profile is never quoted).
Usage: python tools/pmc_summary.py r3_final"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_dispatch_counters(path):
    """{kernel short name: {counter: value of the LAST dispatch, "_solo_ms": its duration in THIS pass}} (counter rows of one
    dispatch are summed over their dimensions, as rocprofv3 lists one row per counter instance).  A counter pass serializes
    the dispatches, so the duration is the kernel's running ALONE — the time its counters belong to."""
    per, span = {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "rsv::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace(", ", " ")  # (no commas inside a CSV field)
            d = per.setdefault(short, {})
            key = (int(r["Dispatch_Id"]), r["Counter_Name"])
            d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                span[(short, int(r["Dispatch_Id"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    for short, d in per.items():
        last = max(k[0] for k in d)
        out[short] = {c: v for (disp, c), v in d.items() if disp == last}
        if (short, last) in span:
            out[short]["_solo_ms"] = span[(short, last)]
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r3_final"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench_n1_65536.json"), os.path.join(dst, f"{tag}_bench_n1_65536.json"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.reader(open(stats)))
    keep = [rows[0]] + [r for r in rows[1:] if "rsv::" in r[0]]
    with open(os.path.join(dst, f"{tag}_kernel_stats_bench65536.csv"), "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
    avg_ms = {r[0].split("(")[0].replace("void ", "").replace(", ", " "): (int(r[1]), float(r[3]) / 1e6) for r in rows[1:] if "rsv::" in r[0]}
    counters = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
        path = glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)[0]
        for k, v in last_dispatch_counters(path).items():
            solo = v.pop("_solo_ms", None)
            counters.setdefault(k, {}).update(v)
            if solo is not None:  # the dispatch's duration in the pass its counters come from
                counters[k]["_solo_ms_" + ("grbm" if name == "GRBM_GUI_ACTIVE" else "sq" if name == "SQ_INSTS_VALU" else name.lower())] = solo
    cols = ["kernel", "calls", "span_ms_concurrent", "solo_ms", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_corrected", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
            "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "eff_clock_GHz", "valu_winst_per_s",
            "valu_issue_frac_of_1.2288e12", "wave_wait_any_frac", "wave_wait_issue_frac", "wave_active_frac",
            "frac_of_perm_mix_issue_ceiling", "note"]
    latest = {"tag": tag, "proofs": 65536, "kernels": {},
              "kernel_sources_sha": open(os.path.join(src, "kernel_sources_sha.txt")).read().strip()}
    with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
        f.write(f"# rocprofv3 summary, {tag} build — `python bench.py --steps 1..2 --warmup 1` (65 536 proofs, 7.71 GB, one MI355X); tools/profile.sh + tools/pmc_summary.py\n")
        f.write(f"# kernel-trace pass: profiles/{tag}_kernel_stats_bench65536.csv; counters from four separate --pmc passes (no trace domains mixed in)\n")
        f.write("# FETCH_SIZE / WRITE_SIZE are KiB as reported; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream, MI355X_MICROARCH.md)\n")
        f.write("# A counter pass serializes the dispatches: a kernel's counters are those of the kernel running ALONE, and so is solo_ms (End - Start of the same dispatch in the SQ pass). "
                "Every rate below divides counters by THAT time: eff_clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / (the dispatch's duration in the GRBM pass); valu_issue_frac = SQ_INSTS_VALU / solo_ms against 1024 SIMDs x 2.4 GHz / 2 cycles\n")
        f.write("# span_ms_concurrent = the kernel's average duration in the kernel-trace pass, where it runs BESIDE the other stream's kernels (what bench.py's HIP events see): a span, not work; "
                "note = 'queued' where the span is mostly waiting for room on the chip (span > 3 x solo)\n")
        f.write("# wave_*_frac: SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (ready but not issued: the VALU is taken by another wave) and SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; they sum to ~1\n")
        f.write("# frac_of_perm_mix_issue_ceiling = SQ_INSTS_VALU x 3.440 cycles-at-2.4-GHz (mean issue cost of the permutation's instruction mix, tools/perm_ceiling.py) / (solo_ms x 1024 SIMDs x 2.4 GHz): meaningful for the permutation-dominated kernels only\n")
        f.write("# kernels on the side stream (k_row_hash, k_query, k_oods, k_qconst; round 3: k_pair_merkle for single-group batches) overlap main-stream kernels: their durations and clocks are not isolated\n")
        f.write(",".join(cols) + "\n")
        for k in sorted(avg_ms, key=lambda k: -avg_ms[k][1] * avg_ms[k][0]):
            c = counters.get(k, {})
            calls, ms = avg_ms[k]
            fetch, write = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            solo = c.get("_solo_ms_sq", 0.0) or ms      # (an old trace without timestamps: fall back to the span)
            solo_g = c.get("_solo_ms_grbm", 0.0) or solo
            clock = gui / 8 / (solo_g * 1e-3) / 1e9 if solo_g > 0 else 0.0  # GRBM_GUI_ACTIVE is reported per XCD (8), summed above
            rate = c.get("SQ_INSTS_VALU", 0.0) / (solo * 1e-3) if solo > 0 else 0.0
            note = "queued" if solo > 0 and ms > 3 * solo else ""
            f.write(",".join(str(x) for x in [k, calls, round(ms, 4), round(solo, 4), int(fetch), int(write), int((2 * fetch + write) * 1024),
                                              int(c.get("SQ_INSTS_VALU", 0)), int(c.get("SQ_INSTS_SALU", 0)), int(c.get("SQ_WAVES", 0)),
                                              int(c.get("SQ_WAVE_CYCLES", 0)), int(c.get("SQ_BUSY_CYCLES", 0)), int(gui),
                                              round(clock, 3), f"{rate:.3e}", round(rate / 1.2288e12, 3),
                                              round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                              round(c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                              round(c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                              round(c.get("SQ_INSTS_VALU", 0) * 3.440 / (solo * 1e-3 * 1024 * 2.4e9), 3) if solo > 0 else 0, note]) + "\n")
            short = k.replace("rsv::", "").split("<")[0]
            if short in latest["kernels"] and " true" in k:
                continue  # a FLOW / side instantiation of a template kernel never replaces the verdict instantiation
            latest["kernels"][short] = {"avg_ms": ms, "solo_ms": round(solo, 4), "hbm_bytes_corrected": int((2 * fetch + write) * 1024),
                                        "SQ_INSTS_VALU": int(c.get("SQ_INSTS_VALU", 0)), "eff_clock_GHz": round(clock, 3),
                                        "valu_issue_frac_solo": round(rate / 1.2288e12, 3)}
    # traffic of one step: every pipeline kernel once (k_cap_top runs twice per step: its trace-tree launch moves about as much
    # as the FRI one counted here, 0.1 GB), against the algorithmic bytes of the bench line
    bench = json.load(open(os.path.join(src, "bench_n1_65536.json")))
    algo = int(bench["roofline"]["algorithmic_bytes_per_launch"])
    pipe = sum(v["hbm_bytes_corrected"] for k, v in latest["kernels"].items() if k not in ("k_permute", "k_emulated", "k_half_permute"))
    latest["pipeline_solo_ms"] = round(sum(v["solo_ms"] for k, v in latest["kernels"].items() if k not in ("k_permute", "k_emulated", "k_half_permute")), 3)
    latest["algorithmic_bytes"] = algo
    latest["pipeline_hbm_bytes"] = pipe
    latest["pipeline_traffic_ratio"] = round(pipe / algo, 3)
    # wave-level VALU instructions of one step (the same kernels): what bench.py prices against the step's wall time — a figure
    # that does not depend on how the overlapping kernels' spans are attributed
    latest["pipeline_valu_insts"] = sum(v["SQ_INSTS_VALU"] for k, v in latest["kernels"].items() if k not in ("k_permute", "k_emulated", "k_half_permute"))
    latest["bench_value_proofs_per_s"] = bench["value"]
    latest["bench_ms_per_step"] = bench["ms_per_step"]
    with open(os.path.join(dst, "pmc_latest.json"), "w") as f:
        json.dump(latest, f, indent=1)
    print("wrote profiles/" + tag + "_*")


if __name__ == "__main__":
    main()
