#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests -q -m gpu -x -k "sixteen_query or ivf_baseline_size or list_centric or pipelined_screen or image" 2>&1 | tail -15 > gpurun_out/r6e/pytest_ivf16.txt
cat gpurun_out/r6e/pytest_ivf16.txt | tail -8
timeout 900 python tools/ivf_width_bench.py 96 > gpurun_out/r6e/ivf_width_bench.txt 2>&1
cat gpurun_out/r6e/ivf_width_bench.txt
