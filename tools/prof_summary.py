#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats csv directory: per-kernel time per step."""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv')
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU kernel time {tot/1e6:.2f} ms; per step ({steps} steps) {tot/1e6/steps:.3f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls'])//steps:>4d}/step avg_us {float(r['AverageNs'])/1e3:8.1f} ms/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} {float(r['Percentage']):5.1f}%")
