"""Prints the kernel sequence of the last few bench steps from a rocprofv3 kernel trace: name, start offset, duration, gap to the previous
kernel (us).  Usage: python scripts/trace_step.py <dir with *_kernel_trace.csv> [n_last_kernels]"""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
