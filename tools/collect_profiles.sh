#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: collects everything profiles/rNN/ holds.  usage: tools/collect_profiles.sh r01
R=${1:-r01}
OUT=gpurun_out/$R
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. the bench line (default flags: N=1, cpu_baseline, roofline)
timeout 600 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"
# 2. kernel trace + stats of the same workload
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --no-cpu-baseline > "$OUT/trace.log" 2>&1
python tools/prof_summary.py "$OUT/trace" 35 40 > "$OUT/bench_summary.txt" 2>&1
cp "$OUT"/trace/bench_kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
cp "$OUT"/trace/bench_domain_stats.csv "$OUT/bench_domain_stats.csv" 2>/dev/null
# 2b. the roofline kernel alone (the probe bench.py times with HIP events): its average duration in this summary is the one to compare
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gtrace" -o g -- python3 tools/gemm_bench.py 30 "NT fc1" > "$OUT/gtrace.log" 2>&1
cp "$OUT"/gtrace/g_kernel_stats.csv "$OUT/gemm_fc1_kernel_stats.csv" 2>/dev/null
# 3. counters of the dominant kernel, one counter set per pass
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" SQ_LDS_BANK_CONFLICT; do
  tag=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$tag" -o g -- python3 tools/gemm_bench.py 5 "NT fc1" > "$OUT/pmc_$tag.log" 2>&1
  cp "$OUT/pmc_$tag/g_counter_collection.csv" "$OUT/gemm_fc1_pmc_$tag.csv" 2>/dev/null
done
# 4. whole-step fabric traffic (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/step_$c" -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/step_$c.log" 2>&1
  python tools/pmc_sum.py "$OUT/step_$c" 5 > "$OUT/step_pmc_$c.txt" 2>&1
done
# 5. retrieval row
timeout 200 python tools/retrieval_bench.py 5 2>/dev/null > "$OUT/retrieval_bench.txt"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rtrace" -o r -- python3 tools/retrieval_bench.py 3 > "$OUT/rtrace.log" 2>&1
python tools/prof_summary.py "$OUT/rtrace" 4 6 > "$OUT/retrieval_summary.txt" 2>&1
rm -rf "$OUT"/trace/*trace.csv "$OUT"/gtrace "$OUT"/rtrace "$OUT"/pmc_* "$OUT"/step_FETCH_SIZE "$OUT"/step_WRITE_SIZE "$OUT"/*.log
ls -la "$OUT"
cat "$OUT/bench_line.json"
