#!/usr/bin/env python
"""Per-launch-group kernel durations from a rocprofv3 --kernel-trace csv directory: consecutive dispatches of one kernel
(same name and grid) form a group (a micro-benchmark loop); prints the median / mean / min duration of each group.
usage: tools/ktrace.py <dir> [min_group]      (KTRACE_BYNAME=1: one group per (kernel, grid) regardless of adjacency -- for loops whose
launches are separated by copy kernels; KTRACE_SPLIT=<substring>: cut the trace into segments at every kernel whose name contains the
substring (a marker the benchmark launches between its legs, e.g. torch.cuda._sleep -> "spin"), group by name inside each segment and
print "segment <i>" headers)"""
import csv, glob, os, statistics, sys
d = sys.argv[1]; min_group = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
split = os.environ.get("KTRACE_SPLIT")
if split:
    segs, cur = [], []
    for r in rows:
        if split in r['Kernel_Name']:
            segs.append(cur); cur = []
        else:
            cur.append(r)
    segs.append(cur)
    for i, seg in enumerate(segs):
        by = {}
        for r in seg:
            key = (r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')))
            by.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        print(f"segment {i}")
        for key, durs in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            if len(durs) < min_group:
                continue
            t = durs[3:] if len(durs) > 6 else durs
            print(f"  {key[0][:90]:90s} grid {key[1]:>7s} n={len(durs):3d} median {statistics.median(t):8.2f} us  mean {statistics.mean(t):8.2f}  min {min(t):8.2f}")
    sys.exit(0)
groups = []
for r in rows:
    key = (r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''))
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if os.environ.get("KTRACE_BYNAME"):
        for g in groups:
            if g[0] == key:
                g[1].append(dur)
                break
        else:
            groups.append((key, [dur]))
    elif groups and groups[-1][0] == key:
        groups[-1][1].append(dur)
    else:
        groups.append((key, [dur]))
for key, durs in groups:
    if len(durs) < min_group:
        continue
    t = durs[3:] if len(durs) > 6 else durs
    print(f"{key[0][:70]:70s} grid {key[1]:>7s} n={len(durs):3d} median {statistics.median(t):8.2f} us  mean {statistics.mean(t):8.2f}  min {min(t):8.2f}")
