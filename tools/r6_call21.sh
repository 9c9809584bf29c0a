#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
( echo "# tools/_exp/adc_retry_cost.py (sel_slack 2 to provoke repeats): before the warm-up of the retry path's operators the first"
  echo "# repeated query of a process cost 301.57 ms (run of the previous commit: batch 10), later ones ~1 ms; with the warm-up:"
  python tools/_exp/adc_retry_cost.py 2>&1 | grep -v amdgpu ) > gpurun_out/r6m/adc_retry_cost.txt
cat gpurun_out/r6m/adc_retry_cost.txt
