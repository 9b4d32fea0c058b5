#!/bin/bash
# Runtime environment A/B on one box: the default bench step (H2D inclusive, 100 steps) under HIP / ROCr settings that change how launches reach the chip.
# usage (GPU box, repo root): bash tools/env_ab.sh > gpurun_out/env_ab.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 100 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | head -1 | grep -o '[0-9.]*$'; }
echo "# ms per step, bench.py --steps 100 --warmup 10 (H2D inclusive), alternating passes on one box"
for pass in 1 2 3; do
  echo "pass $pass"
  echo "  default                      $(timeout 200 $B 2>/dev/null | ms)"
  echo "  HIP_FORCE_DEV_KERNARG=1      $(HIP_FORCE_DEV_KERNARG=1 timeout 200 $B 2>/dev/null | ms)"
  echo "  HIP_FORCE_DEV_KERNARG=0      $(HIP_FORCE_DEV_KERNARG=0 timeout 200 $B 2>/dev/null | ms)"
  echo "  HSA_ENABLE_INTERRUPT=0       $(HSA_ENABLE_INTERRUPT=0 timeout 200 $B 2>/dev/null | ms)"
  echo "  HSA_ENABLE_SDMA=0            $(HSA_ENABLE_SDMA=0 timeout 200 $B 2>/dev/null | ms)"
  echo "  HSA_NO_SCRATCH_RECLAIM=1     $(HSA_NO_SCRATCH_RECLAIM=1 timeout 200 $B 2>/dev/null | ms)"
  echo "  GPU_STREAMOPS_CP_WAIT=1      $(GPU_STREAMOPS_CP_WAIT=1 timeout 200 $B 2>/dev/null | ms)"
done
