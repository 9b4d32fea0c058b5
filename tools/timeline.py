#!/usr/bin/env python
"""Concurrency analysis of a rocprofv3 kernel trace: tools/timeline.py <kernel_trace.csv> [steps]
Takes the last `steps` occurrences of k_adamw as step boundaries and reports, over that window: wall time per step, time with
0/1/2/3/4+ kernels in flight, per-queue busy time, the estimated CU-slot occupancy (sum over kernels of min(1, workgroups*wg_share)),
and the largest idle gaps with the kernels around them."""
import csv, sys, collections
fn = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]),
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])),
                 int(r["LDS_Block_Size"])))
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith("k_adamw")]
i0, i1 = ad[-steps - 1], ad[-1]
t0, t1 = rows[i0][1], rows[i1][1]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
wall = (t1 - t0) / 1e3
print(f"window: {steps} steps, {wall/steps:.1f} us per step, {len(win)/steps:.0f} kernels per step")
ev = []
for s, e, *_ in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    hist[min(cur, 4)] += t - last; last = t; cur += d
hist[min(cur, 4)] += t1 - last
tot = sum(hist.values())
print("kernels in flight:", "  ".join(f"{k}{'+' if k == 4 else ''}: {100*v/tot:.1f}%" for k, v in sorted(hist.items())))
q = collections.Counter()
for s, e, n, qid, *_ in win:
    q[qid] += e - s
print("busy time per queue (us/step):", {k: round(v / 1e3 / steps, 1) for k, v in sorted(q.items())})
# CU-slot occupancy estimate: a kernel with W workgroups occupies min(W, 512)/512 of the chip's 2-per-CU slots (1-per-CU kernels count double)
occ = 0.0
for s, e, n, qid, wgs, lds in win:
    per_cu = 1 if lds > 65536 else 2
    occ += (e - s) * min(1.0, wgs / (256.0 * per_cu))
print(f"workgroup-slot occupancy (time-weighted, all kernels): {occ/(t1-t0):.2f} chips")
# idle gaps
gaps = []
cur = 0; last = None
for t, d in ev:
    if cur == 0 and last is not None and d == 1 and t - last > 2000: gaps.append((t - last, last, t))
    cur += d
    if cur == 0: last = t
gaps.sort(reverse=True)
print(f"idle (no kernel running): {hist[0]/1e3/steps:.1f} us per step; largest gaps:")
for g, a, b in gaps[:6]:
    before = [r[2][:50] for r in win if abs(r[1] - a) < 50]; after = [r[2][:50] for r in win if abs(r[0] - b) < 50]
    print(f"   {g/1e3:7.1f} us  after {before[:1]}  before {after[:1]}")
