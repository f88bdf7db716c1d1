"""Aggregates the rocprofv3 output of scripts/profile_bench.sh into small CSVs (kernel stats, PMC
counters per kernel) under gpurun_out/prof_<tag>/summary/ - copy them to profiles/ to commit."""
import csv, glob, os, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(rows)

# per-launch durations of the dominant kernel from the kernel trace, in launch order, and their average over the
# launches of bench.py's timed region (after the pre-warm and warm-up passes, before the greedy extra) - the number the
# HIP-event average of bench.py must agree with; the stats file's own average also contains the clock-ramp launches
trace = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    PRE, WARM, STEPS = 8, 3, 20  # bench.py: extra["pre_warm_steps"], --warmup, --steps of scripts/profile_bench.sh
    try:  # round 4: bench.py pre-warms by time, the count is in its JSON line (first launch of a run = the factorisation's mean pass)
        import json as _json

        _line = _json.load(open(os.path.join(out, "bench_line.json")))
        PRE, WARM, STEPS = int(_line["extra"]["pre_warm_steps"]), int(_line["warmup"]), int(_line["steps"])
    except Exception:  # noqa: BLE001
        pass
    rows = [r for r in csv.DictReader(open(trace[0])) if "posterior_kernel" in r["Kernel_Name"]]
    if any("qlognehvi" in r["Kernel_Name"] for r in csv.DictReader(open(trace[0]))):  # cfg5: three variance launches per step
        rows = [r for r in csv.DictReader(open(trace[0])) if "qlognehvi_lin_kernel" in r["Kernel_Name"] or "qlognehvi_kernel" in r["Kernel_Name"]]
    coop_rows = [r for r in rows if "bbh_coop" in r["Kernel_Name"]]  # (the windowed form's launches in a cooperative run are the
    if len(coop_rows) >= STEPS:                                          # mean-only passes of the set-up and of the greedy extra)
        rows = coop_rows
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    timed = dur[PRE + WARM : PRE + WARM + STEPS]
    with open(os.path.join(dst, f"{tag}_bench_posterior_launches.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["launch", "region", "duration_ms"])
        for i, t in enumerate(dur):
            region = ("pre-warm" if i < PRE else "warm-up" if i < PRE + WARM else "timed" if i < PRE + WARM + STEPS else
                      "instrumented (every family's events)" if i < PRE + WARM + 2 * STEPS + 1 else "extras (greedy batch, end to end)")
            w.writerow([i, region, f"{t:.4f}"])
        if timed:
            w.writerow(["timed-region average", len(timed), f"{sum(timed) / len(timed):.4f}"])

acc = defaultdict(list)
# L2 <-> fabric bytes per launch of the dominant kernels: FETCH_SIZE (KiB; doubled - gfx950 reports half the bytes of
# coalesced streaming reads, MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB), from their separate PMC passes
def _traffic(acc_):
    per = defaultdict(dict)
    for (ps, k, c), v in acc_.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE") and ("posterior_kernel" in k or "columns_kernel" in k or "qlognehvi" in k):
            per[k.split("(")[0]][c] = sum(v) / len(v)
    return {k: {"fetch_size_kib": d.get("FETCH_SIZE"), "write_size_kib": d.get("WRITE_SIZE"),
                "hbm_bytes_per_launch": int((2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024)} for k, d in per.items()}


for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[(os.path.basename(d), r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_bench_pmc_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "dispatches", "max_per_dispatch", "mean_per_dispatch"])
    for (ps, k, c), v in sorted(acc.items()):
        w.writerow([ps, k, c, len(v), max(v), sum(v) / len(v)])
import json

with open(os.path.join(dst, f"{tag}_traffic.json"), "w") as f:
    json.dump(_traffic(acc), f, indent=1)

# ---- pipe-level issue fraction of the VALU-bound kernels ------------------------------------------------------
# issue_frac = (VALU wave-instructions x mean cycles of the kernel's static class mix + MFMA x 64) / (SIMDs x clock x duration), with
# SQ_INSTS_VALU taken to INCLUDE the MFMA instructions (they are subtracted), the clock from GRBM_GUI_ACTIVE (summed over the 8 XCDs)
# over the traced duration of the same dispatches, and 1024 SIMDs.
mix_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r04_isa_class_mix.json")
if os.path.exists(mix_file):
    mix = json.load(open(mix_file))
    cnt, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for d in sorted(glob.glob(os.path.join(out, "pmcg_*"))):
        if not os.path.isdir(d):
            continue
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    issue = {}
    for shown, rec in mix["kernels"].items():
        for kname, c in cnt.items():
            if shown not in kname or "SQ_INSTS_VALU" not in c:
                continue
            mean = lambda v: sum(v) / len(v)
            valu_all, mfma = mean(c["SQ_INSTS_VALU"]), mean(c.get("SQ_INSTS_MFMA", [0.0]))
            t_ns = mean(dur[kname]) if dur.get(kname) else None
            if not t_ns:
                continue
            clock = mean(c["GRBM_GUI_ACTIVE"]) / 8.0 / (t_ns * 1e-9) if "GRBM_GUI_ACTIVE" in c else 2.4e9
            cyc = (valu_all - mfma) * rec["mean_cycles_per_valu_instruction"] + mfma * 64.0
            issue[shown] = {"dispatches": len(c["SQ_INSTS_VALU"]), "valu_wave_instructions_excl_mfma": valu_all - mfma, "mfma_wave_instructions": mfma,
                            "mean_cycles_per_valu_instruction_static_mix": rec["mean_cycles_per_valu_instruction"],
                            "duration_us_under_pmc": t_ns / 1e3, "clock_ghz_from_grbm_gui_active": clock / 1e9,
                            "issue_frac": cyc / (1024.0 * clock * t_ns * 1e-9)}
    with open(os.path.join(dst, f"{tag}_issue.json"), "w") as f:
        json.dump({"_comment": "pipe-level issue fraction of the VALU-bound kernels: see scripts/profile_summarize.py and profiles/r04_isa_class_mix.json",
                   "kernels": issue}, f, indent=1)
line = os.path.join(out, "bench_line.json")
if os.path.exists(line):
    try:
        rec = json.loads(open(line).read())
        json.dump(rec, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
    except Exception:  # noqa: BLE001
        pass
print("summaries in", dst, os.listdir(dst))
