#!/usr/bin/env python
"""Join tools/vendor_yardstick.py's printed segment labels with `KTRACE_SPLIT=spin tools/ktrace.py`'s per-segment kernel durations:
one line per (shape, rows) with OUR kernel's and the VENDOR kernel's median device-side duration and the vendor kernel's name (macro tile).
usage: tools/vendor_yardstick_join.py <events.txt> <segments.txt>"""
import re, sys
labels = {}
for l in open(sys.argv[1]):
    m = re.match(r"#\s+segment (\d+): (ours|vendor)\s+(.*) rows (\d+)", l)
    if m: labels[int(m.group(1))] = (m.group(2), m.group(3).strip(), int(m.group(4)))
segs, cur = {}, None
for l in open(sys.argv[2]):
    if l.startswith("segment"):
        cur = int(l.split()[1]); segs[cur] = []
    elif cur is not None and "median" in l:
        name = l[:92].strip()
        med = float(re.search(r"median\s+([0-9.]+)", l).group(1)); n = int(re.search(r"n=\s*(\d+)", l).group(1))
        segs[cur].append((name, n, med))
SKIP = ("elementwise", "fill", "copy", "memset", "reduce_kernel", "distribution", "spin")
rows = {}
for i, (side, shape, M) in labels.items():
    ks = [k for k in segs.get(i, []) if not any(s in k[0].lower() for s in SKIP)]
    if not ks: continue
    # the GEMM of the leg: the kernel with the most launches, ties to the longest (our dW leg also launches a bias-reduction kernel)
    ks.sort(key=lambda k: (-k[1], -k[2]))
    main = ks[0]
    extra = sum(k[2] for k in ks[1:] if k[1] >= main[1] - 3)
    rows.setdefault((shape, M), {})[side] = (main[0], main[2], extra)
FL = {"qkv fwd": (1152, 384), "proj fwd": (384, 384), "fc1 fwd": (1536, 384), "fc2 fwd": (384, 1536), "fc2 dX": (1536, 384), "fc1 dX": (384, 1536),
      "proj dX": (384, 384), "qkv dX": (384, 1152), "dW qkv": (1152, 384), "dW proj": (384, 384), "dW fc1": (1536, 384), "dW fc2": (384, 1536)}
print(f"# {'shape':9s} {'rows':>6s} | {'ours us':>8s} {'TF/s':>6s} | {'vendor us':>9s} {'TF/s':>6s} | ours/vendor | vendor kernel (hipBLASLt / Tensile solution: MT = macro tile M x N x depth-K)")
for (shape, M), d in rows.items():
    if "ours" not in d or "vendor" not in d: continue
    n, k = FL[shape]; fl = 2.0 * M * n * k
    o, v = d["ours"], d["vendor"]
    ot = o[1] + o[2]
    mt = re.search(r"MT\d+x\d+x\d+", v[0])
    print(f"  {shape:9s} {M:6d} | {ot:8.2f} {fl / ot / 1e6:6.0f} | {v[1]:9.2f} {fl / v[1] / 1e6:6.0f} | {ot / v[1]:11.2f} | {mt.group(0) if mt else v[0][:40]}"
          + (f"   (ours = {o[1]:.1f} GEMM + {o[2]:.1f} bias-gradient / reduction kernels)" if o[2] else ""))
