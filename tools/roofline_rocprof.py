#!/usr/bin/env python
"""Average device-side duration of the roofline kernel from a rocprofv3 --kernel-trace --stats run of tools/gemm_bench.py at the roofline shape
-> profiles/rNN/roofline_rocprof.json (bench.py prints roofline.frac from HIP events and from this side by side).
usage: tools/roofline_rocprof.py OUTDIR trace_dir"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out_dir, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_gemm_mfma" in r["Name"]]
r = max(rows, key=lambda r: int(r["Calls"]))
rec = dict(kernel=r["Name"].split("(")[0], shape=[bench.ROOF_KIND, bench.ROOF_M, bench.ROOF_N, bench.ROOF_K], calls=int(r["Calls"]),
           us_per_launch=round(float(r["AverageNs"]) / 1e3, 3), source=f"profiles/{os.path.basename(out_dir)}/roofline_kernel_stats.csv",
           source_stamp=bench.kernel_source_stamp())
json.dump(rec, open(os.path.join(out_dir, "roofline_rocprof.json"), "w"), indent=1)
print(json.dumps(rec))
