#!/bin/bash
# A/B of two builds of the library on ONE GPU box (boxes differ by +-3 %): tools/ab_bench.sh base.so [rounds]
# usage from the build container:  git stash; python -m fedcola_amd.build; cp fedcola_amd/libfedcola_hip.so fedcola_amd/libfc_base.so; git stash pop; python -m fedcola_amd.build
#                                  gpurun -- 'bash tools/ab_bench.sh fedcola_amd/libfc_base.so'
BASE=${1:-fedcola_amd/libfc_base.so}; R=${2:-3}
for i in $(seq $R); do
  a=$(FC_LIB_PATH=$PWD/$BASE python bench.py --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 100 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(python bench.py --no-cpu-baseline --no-roofline --no-dropout-line --no-extra-legs --steps 100 --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "round $i: base $a   new $b"
done
