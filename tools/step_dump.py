#!/usr/bin/env python
"""One client step from a rocprofv3 kernel trace, kernel by kernel: tools/step_dump.py <kernel_trace.csv> [which_from_end]
(start offset within the step, queue, duration, name).  Step boundaries = the last k_adamw launch of each step."""
import csv, sys, re
fn = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith("k_adamw")]
# a step ends with a run of adamw launches: take the last index of each run
ends = [i for k, i in enumerate(ad) if k + 1 == len(ad) or rows[ad[k + 1]][0] - rows[i][1] > 200000]
i0, i1 = ends[-back - 1], ends[-back]
t0 = rows[i0][1]
qs = sorted({r[3] for r in rows[i0 + 1:i1 + 1]})
print(f"step of {(rows[i1][1]-t0)/1e3:.1f} us, {i1-i0} kernels, queues {qs}")
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("unsigned short", "bf16").replace("void ", "")
    return n[:44]
for s, e, n, q, wgs in rows[i0 + 1:i1 + 1]:
    print(f"{(s-t0)/1e3:8.1f} {'    ' * qs.index(q)}q{qs.index(q)} {(e-s)/1e3:7.1f}  {short(n)} [{wgs}]")
