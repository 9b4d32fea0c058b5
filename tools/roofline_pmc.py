#!/usr/bin/env python
"""Per-launch PMC means of the roofline kernel -> profiles/rNN/roofline_pmc.json (read by bench.py, which refuses it when the
kernel sources or the shape have changed since).  usage: tools/roofline_pmc.py OUTDIR counter_dir [counter_dir ...]
Each counter_dir holds one rocprofv3 --pmc pass (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out_dir, dirs = sys.argv[1], sys.argv[2:]
KERNEL = "k_gemm_mfma"          # the run launches only the roofline shape (GEMM_SHAPES), so any k_gemm_mfma dispatch is it
vals = {}
rec_name = KERNEL
for d in dirs:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(fn)):
            if KERNEL not in r["Kernel_Name"]:
                continue
            rec_name = r["Kernel_Name"].split("(")[0]
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            v = v[3:] if len(v) > 6 else v            # skip warm-up launches
            vals[k] = sum(v) / len(v)
rec = dict(kernel=rec_name + " (NN dX GEMM) at the roofline shape of bench.py", shape=[bench.ROOF_KIND, bench.ROOF_M, bench.ROOF_N, bench.ROOF_K],
           source_stamp=bench.kernel_source_stamp(), source_stamp_covers=list(bench.ROOF_SOURCES))
for k, v in vals.items():
    rec[k + "_per_launch"] = v
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # rocprofv3 reports KB; gfx950 tallies a 128-B read request as 64 B (MI355X_MICROARCH.md, HBM): FETCH_SIZE x 2
    rec["hbm_bytes_per_launch"] = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    M, N, K = bench.ROOF_M, bench.ROOF_N, bench.ROOF_K
    rec["algorithmic_bytes_per_launch"] = 2 * (M * K + N * K + M * N)
    rec["note"] = ("separate rocprofv3 --pmc passes over tools/gemm_bench.py at the roofline shape; FETCH_SIZE doubled (gfx950 counts 128-B requests "
                   "as 64 B), WRITE_SIZE as reported")
json.dump(rec, open(os.path.join(out_dir, "roofline_pmc.json"), "w"), indent=1)
print(json.dumps(rec))
