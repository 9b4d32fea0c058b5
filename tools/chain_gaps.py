#!/usr/bin/env python
"""Per-queue view of a rocprofv3 --kernel-trace csv of bench.py: how much of a step each stream spends running kernels and how much
between them.  tools/chain_gaps.py <kernel_trace.csv> [steps]
A step of the client is dependency chains of ~200 kernels each (three image micro-batch chains, a text tower, the weight-gradient
stream): if a chain's stream is idle between two of its kernels, the step waits.  Reports, per queue and step: kernels, busy time, time
between consecutive kernels (end -> next start) split into short gaps (< 20 us: launch / dependency latency) and long ones (waiting for
another stream), and the distribution of the short gaps."""
import csv, sys, collections
fn = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith("k_adamw")]
marks = sorted(set(rows[i][1] for i in ad))
# one k_adamw group per step: take the last launch of each step (launches closer than 1 ms belong to one step)
ends = []
for t in marks:
    if ends and t - ends[-1] < 1_000_000: ends[-1] = t
    else: ends.append(t)
t0, t1 = ends[-steps - 1], ends[-1]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
print(f"window: {steps} steps, {(t1 - t0) / 1e3 / steps:.1f} us per step (traced), {len(win) / steps:.0f} kernels per step")
byq = collections.defaultdict(list)
for r in win: byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    rs.sort()
    busy = sum(e - s for s, e, *_ in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    short = sorted(g for g in gaps if 0 <= g < 20000); long_ = [g for g in gaps if g >= 20000]; neg = [g for g in gaps if g < 0]
    if not short: continue
    med = short[len(short) // 2]; p90 = short[len(short) * 9 // 10]
    print(f"queue {q}: {len(rs) / steps:6.1f} kernels/step  busy {busy / 1e3 / steps:7.1f} us  short gaps {sum(short) / 1e3 / steps:7.1f} us "
          f"(n {len(short) / steps:.0f}, median {med / 1e3:.2f}, p90 {p90 / 1e3:.2f} us)  long gaps {sum(long_) / 1e3 / steps:7.1f} us (n {len(long_) / steps:.1f})  overlapping {len(neg) / steps:.0f}")
    # what the short gaps sit behind: average gap after each kernel family
    fam = collections.defaultdict(list)
    for i, g in enumerate(gaps):
        if 0 <= g < 20000: fam[rs[i][2].split("(")[0][:48]].append(g)
    top = sorted(fam.items(), key=lambda kv: -sum(kv[1]))[:6]
    for name, gs in top:
        print(f"      after {name:48s} n/step {len(gs) / steps:5.1f}  mean {sum(gs) / len(gs) / 1e3:5.2f} us  total {sum(gs) / 1e3 / steps:6.1f} us/step")
