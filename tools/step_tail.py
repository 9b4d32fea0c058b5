#!/usr/bin/env python
"""The last kernels of a client step, from a rocprofv3 --kernel-trace csv: tools/step_tail.py <dir> [kernels] [step_from_end]
Prints, for one steady-state step (boundary: k_adamw_chunks ... next k_adamw_chunks), the final `kernels` launches with queue, start and end
relative to the step's last kernel end, duration and grid -- what the exposed tail of the step is made of."""
import csv, glob, sys
d = sys.argv[1]; nk = int(sys.argv[2]) if len(sys.argv) > 2 else 30; back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in csv.DictReader(open(f))]
rows.sort()
ends = sorted(r[1] for r in rows if r[2].startswith("k_gemm_dw_spec") or "k_gemm_dw_spec" in r[2])
# a step ends with its last weight-gradient launch: take the dW launch ends that are followed by a gap to the next dW launch's start
dw = [r for r in rows if "k_gemm_dw_spec" in r[2]]
last_of_step = [dw[i] for i in range(len(dw)) if (i + 1) % 4 == 0]
t_end = last_of_step[-back][1]
t_prev = last_of_step[-back - 1][1]
win = [r for r in rows if r[1] <= t_end + 200000 and r[0] >= t_prev]
step = [r for r in win if r[0] < t_end + 100000]
print(f"# step window {(t_end - t_prev) / 1e3:.0f} us between the ends of two consecutive last weight-gradient launches; times in us relative to this step's one")
for s, e, n, q, wg in sorted(step, key=lambda r: r[1])[-nk:]:
    print(f"q{q:<3d} start {(s - t_end) / 1e3:9.1f} end {(e - t_end) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} wgs {wg:6d}  {n[:80]}")
# ---- the middle of the step: from the first head kernel of the forward to the first attention backward (the serial section around the loss)
if len(sys.argv) > 4 and sys.argv[4] == "mid":
    st = sorted(step)
    i0 = next(i for i, r in enumerate(st) if "k_head_fwd" in r[2])
    i1 = next(i for i, r in enumerate(st) if "k_attn_bwd" in r[2])
    t0 = st[i0][0]
    print(f"# middle of the step: {(st[i1][0] - t0) / 1e3:.0f} us from the first head kernel to the first attention backward; us relative to the former")
    for s, e, n, q, wg in st[max(0, i0 - 6):i1 + 1]:
        print(f"q{q:<3d} start {(s - t0) / 1e3:9.1f} end {(e - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} wgs {wg:6d}  {n[:80]}")
# ---- per-queue progress: end time of every image-tower attention kernel (one per layer and chain), forward then backward
if len(sys.argv) > 4 and sys.argv[4] == "progress":
    st = sorted(step)
    t0 = st[0][0]
    for tag, pat in (("forward", "k_attn_fwd_mfma<14"), ("backward", "k_attn_bwd<14")):
        per = {}
        for s, e, n, q, wg in st:
            if pat in n: per.setdefault(q, []).append((e - t0) / 1e3)
        print(f"# {tag}: end of the image tower's attention kernel per layer, us after the step's first kernel, one row per queue")
        for q, v in sorted(per.items()):
            print(f"q{q}: " + " ".join(f"{x:7.0f}" for x in v))
