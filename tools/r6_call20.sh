#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
python tools/ivf_slack_bench.py 6 5 4 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r6m/ivf_slack.txt
cat gpurun_out/r6m/ivf_slack.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "adc or search or robust" 2>&1 | tail -3
