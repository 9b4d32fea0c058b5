#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: collects everything profiles/rNN/ holds.  usage: tools/collect_profiles.sh r02
R=${1:-r06}
OUT=gpurun_out/$R
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOF=$(python -c "import bench; print(','.join(map(str, (bench.ROOF_KIND, bench.ROOF_M, bench.ROOF_N, bench.ROOF_K))))")   # the NN dX GEMM at its in-model shape
# 3. counters of the roofline kernel first (bench.py reads roofline_pmc.json), one counter set per pass, kernel-trace only
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" SQ_LDS_BANK_CONFLICT; do
  tag=$(echo $c | tr ' ' '_')
  GEMM_SHAPES="$ROOF" timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/pmc_$tag" -o g -- python3 tools/gemm_bench.py 8 > "$OUT/pmc_$tag.log" 2>&1
  cp /tmp/pmc_$tag/g_counter_collection.csv "$OUT/roofline_pmc_$tag.csv" 2>/dev/null || find /tmp/pmc_$tag -name "*counter_collection.csv" -exec cp {} "$OUT/roofline_pmc_$tag.csv" \;
done
mkdir -p profiles/$R
python tools/roofline_pmc.py profiles/$R /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE /tmp/pmc_SQ_LDS_BANK_CONFLICT > "$OUT/roofline_pmc_print.txt" 2>&1
cp profiles/$R/roofline_pmc.json "$OUT/roofline_pmc.json"
# 2b. the roofline kernel alone: device-side duration to compare with roofline.us_per_launch (HIP events inside bench.py)
GEMM_SHAPES="$ROOF" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gtrace -o g -- python3 tools/gemm_bench.py 30 > "$OUT/gtrace.log" 2>&1
find /tmp/gtrace -name "*kernel_stats.csv" -exec cp {} "$OUT/roofline_kernel_stats.csv" \;
python tools/roofline_rocprof.py profiles/$R /tmp/gtrace > "$OUT/roofline_rocprof_print.txt" 2>&1
cp profiles/$R/roofline_rocprof.json "$OUT/roofline_rocprof.json"
python tools/ktrace.py /tmp/gtrace > "$OUT/roofline_kernel_trace.txt" 2>&1
# 1. the bench line (default flags: N=1, cpu_baseline, roofline with the PMC traffic just measured, dropout-0.1 line)
timeout 900 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"
# 1b. the multi-rank path on this one device (two ranks on cuda:0, gloo for the all-reduce: RCCL refuses two ranks per device)
FC_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 30 --warmup 5 --no-roofline > "$OUT/bench_2ranks_one_device.json" 2> "$OUT/bench_2ranks_stderr.txt"
# 2. kernel trace + stats of the same workload (35 steps)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace -o bench -- python3 bench.py --no-cpu-baseline --no-dropout-line --no-roofline --no-extra-legs --device-resident --steps 30 --warmup 5 > "$OUT/trace.log" 2>&1
python tools/prof_summary.py /tmp/trace 35 40 > "$OUT/bench_summary.txt" 2>&1
find /tmp/trace -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
# 2c. every GEMM shape of a layer + the non-GEMM kernels, stand-alone (device-side durations)
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gall -o g -- python3 tools/gemm_bench.py 20 > /dev/null 2>&1; python tools/ktrace.py /tmp/gall > "$OUT/gemm_shapes_ktrace.txt" 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kb -o g -- python3 tools/kernel_bench.py 20 > /dev/null 2>&1; KTRACE_BYNAME=1 python tools/ktrace.py /tmp/kb > "$OUT/kernels_ktrace.txt" 2>&1
# 2c'. round 4: per kernel family in-step rates from the trace of 2., and the cold-protocol kernel times at the three row counts of the model
python tools/in_situ_roofline.py /tmp/trace 35 > "$OUT/in_situ_roofline.txt" 2>&1
for M in 12608 4334 2048; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/cold$M -o g -- python3 tools/cold_bench.py $M 24 > /dev/null 2>&1
  KTRACE_BYNAME=1 python tools/ktrace.py /tmp/cold$M | grep -v "at::native" > "$OUT/cold_ktrace_$M.txt" 2>&1
done
# 2d. GPU time of the step's phases, no profiler (tools build)
timeout 200 python tools/step_phases.py 40 2>/dev/null > "$OUT/step_phases.txt"
# 4. whole-step fabric traffic (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/step_$c" -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --device-resident > "$OUT/step_$c.log" 2>&1
  python tools/pmc_sum.py "/tmp/step_$c" 5 > "$OUT/step_pmc_$c.txt" 2>&1
done
# 5. widened rows
timeout 200 python tools/retrieval_bench.py 5 2>/dev/null > "$OUT/retrieval_bench.txt"
timeout 300 python tools/cream_bench.py 2>/dev/null > "$OUT/cream_bench.txt"
timeout 300 python tools/loader_bench.py 1024 2>/dev/null > "$OUT/loader_bench.txt"
timeout 300 python tools/vitb_step.py 64 50 2>/dev/null > "$OUT/vitb_step.txt"
timeout 300 python tools/unimodal_step.py 50 2>/dev/null > "$OUT/unimodal_step.txt"
# 6. round-3 records: aggregation kernels, configs[0]'s model, where a client round's wall time goes, and the schedules that lost
timeout 300 python tools/aggregate_bench.py 8 10 2>/dev/null > "$OUT/aggregate_bench.txt"
timeout 200 python tools/tiny_step.py 50 2>/dev/null > "$OUT/tiny_step.txt"
timeout 200 python tools/client_round_trace.py 2>/dev/null > "$OUT/client_round_trace.txt"
if [ -f fedcola_amd/libfedcola_hip_probes.so ]; then
  B2="python bench.py --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 100 --warmup 10 --device-resident"
  ms() { grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; }
  { echo "# ms per step, tools build, same box, two passes each (bench.py --steps 100 --warmup 10)"
    for i in 1 2; do
      echo "streams (product default: 3 image chains forward and backward, text on dW)  $(FC_PROBES_LIB=1 timeout 200 $B2 2>/dev/null | ms)"
      echo "streams with 2 backward chains 57:43, text on its own stream (FC_BWD_CHAINS=2) $(FC_PROBES_LIB=1 FC_BWD_CHAINS=2 timeout 200 $B2 2>/dev/null | ms)"
      echo "streams + per-layer HIP graphs (FC_GRAPHS=1)                            $(FC_PROBES_LIB=1 FC_GRAPHS=1 timeout 200 $B2 2>/dev/null | ms)"
      echo "one chain of grouped launches (FC_SCHEDULE=chain)                       $(FC_PROBES_LIB=1 FC_SCHEDULE=chain timeout 200 $B2 2>/dev/null | ms)"
      echo "two chains of grouped launches (FC_SCHEDULE=chain2)                     $(FC_PROBES_LIB=1 FC_SCHEDULE=chain2 timeout 200 $B2 2>/dev/null | ms)"
      echo "2 forward chains (FC_FWD_CHAINS=2)                                      $(FC_PROBES_LIB=1 FC_FWD_CHAINS=2 timeout 200 $B2 2>/dev/null | ms)"
      echo "2 backward chains, text tower last (FC_BWD_CHAINS=2 FC_TEXT_FIRST=0)    $(FC_PROBES_LIB=1 FC_BWD_CHAINS=2 FC_TEXT_FIRST=0 timeout 200 $B2 2>/dev/null | ms)"
    done; } > "$OUT/schedules_ab.txt"
  for sch in streams chain; do FC_PROBES_LIB=1 FC_SCHEDULE=$sch timeout 200 python tools/step_phases.py 40 2>/dev/null > "$OUT/step_phases_$sch.txt"; done
  bash tools/ablate_ab.sh > "$OUT/ablate.txt" 2>&1
  { for i in 1 2; do echo "FC_MICROBATCH=1 $(FC_PROBES_LIB=1 FC_MICROBATCH=1 $B2 2>/dev/null | ms)"; done; } > "$OUT/microbatch1.txt"
fi
# 7. round 4: the fp32 mode on the matrix cores (split-operand GEMMs, fp32 MFMA attention)
timeout 200 python tools/x3_accuracy.py 2>/dev/null | grep kind > "$OUT/x3_accuracy_now.txt"
timeout 200 python tools/x3_bench.py 2>/dev/null | grep -E "gemm|attention" > "$OUT/x3_bench.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr32 -o g -- python3 bench.py --precision fp32 --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 5 --warmup 2 > "$OUT/tr32.log" 2>&1
python tools/stats_top.py /tmp/tr32 7 > "$OUT/fp32_step_kernels.txt" 2>&1
rm -f "$OUT"/*.log
ls -la "$OUT"
cat "$OUT/bench_line.json"
