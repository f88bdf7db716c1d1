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

acc = defaultdict(list)
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
print("summaries in", dst, os.listdir(dst))
