#!/usr/bin/env python
"""How many kernels run at once, from a rocprofv3 --kernel-trace csv of bench.py: share of the steady-state wall time with 0, 1, 2, ...
kernels in flight, and the kernel time spent at each depth.  usage: tools/concurrency.py <dir> [first_step] [n_steps]
(steps are delimited by the first k_patchify launch of each step: 2 launches per step, one per micro-batch chain)
Caveat: kernel tracing slows the host side of every launch; the traced step (7.5 ms against 5.0 ms untraced) is launch-bound and shows
LESS overlap than the real one (57 % of the traced time with a single kernel in flight).  Use tools/step_phases.py for untraced timing."""
import csv, glob, sys
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 8; nst = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
marks = [s for s, e, n in ev if 'k_patchify' in n][::2]
t0, t1 = marks[first], marks[first + nst]
pts = []
for s, e, n in ev:
    if e <= t0 or s >= t1: continue
    pts.append((max(s, t0), 1)); pts.append((min(e, t1), -1))
pts.sort()
depth, last, hist = 0, t0, {}
for t, dlt in pts:
    hist[depth] = hist.get(depth, 0) + (t - last); last = t; depth += dlt
hist[depth] = hist.get(depth, 0) + (t1 - last)
tot = t1 - t0
print(f"{nst} steps, {tot / nst / 1e6:.3f} ms per step (traced)")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {100.0 * hist[k] / tot:5.1f} % of the time   ({hist[k] / nst / 1e6:.3f} ms per step)")
