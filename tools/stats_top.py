#!/usr/bin/env python
"""Top kernels of a rocprofv3 --kernel-trace --stats run: tools/stats_top.py <dir or kernel_stats.csv> <steps in the run> [rows]"""
import csv, glob, os, sys
src = sys.argv[1]
f = src if os.path.isfile(src) else glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# kernel time per step summed over concurrent streams: {tot / 1e6 / steps:.2f} ms ({steps} steps in the run)")
for r in rows[:n]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d}  {float(r["TotalDurationNs"]) / 1e6 / steps:8.2f} ms/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
