#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
python tools/adc_slack_bench.py 6 4 3 2 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r6m/adc_slack.txt
cat gpurun_out/r6m/adc_slack.txt
