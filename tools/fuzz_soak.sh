#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-soak}; mkdir -p $O
( echo "# fuzz soak on HEAD (tools/fuzz_soak.sh [tag]): fuzz_ivf (every trial: lists / lists8 / lists16 / scan), fuzz_adc, fuzz_nearest"
  for seed in 11 12 13; do timeout 1500 python tools/fuzz_ivf.py $seed 100 2>&1 | grep -v amdgpu.ids | tail -3; done
  for seed in 21 22; do timeout 1500 python tools/fuzz_adc.py $seed 60 2>&1 | grep -v amdgpu.ids | tail -2; done
  timeout 900 python tools/fuzz_nearest.py 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/soak.txt 2>&1
cat $O/soak.txt
