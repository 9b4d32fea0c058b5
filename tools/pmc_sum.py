#!/usr/bin/env python
"""Sum a rocprofv3 --pmc counter_collection csv per kernel name: tools/pmc_sum.py DIR [steps]"""
import csv, glob, sys, collections
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:70]
        tot[r['Counter_Name']][k] += float(r['Counter_Value']); calls[(r['Counter_Name'], k)] += 1
for c, per in tot.items():
    s = sum(per.values())
    print(f"== {c}: total {s:.4g}  per step {s/steps:.4g}")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:16]:
        print(f"   {k:70s} {v/steps:14.4g}/step  calls {calls[(c,k)]/steps:.0f}")
