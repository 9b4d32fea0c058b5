#!/bin/bash
# Marginal wall-clock cost of each kernel family in the client step (tools build: FC_ABLATE skips that family's launches; results are
# wrong while set, only the time counts).  Run on the GPU box: gpurun -- 'bash tools/ablate_ab.sh > gpurun_out/ablate.txt'
export FC_PROBES_LIB=1
B="python bench.py --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 100 --warmup 10 --device-resident"
ms() { grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; }
echo "# ms per step (bench.py --steps 100 --warmup 10, tools build, one box), two passes"
for i in 1 2; do
  for a in none ln attn adamw dw txt gemm "${@}"; do
    echo "FC_ABLATE=$a  $(FC_ABLATE=$a timeout 200 $B 2>/dev/null | ms)"
  done
done
