#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
timeout 1200 python -m pytest tests -q -m gpu -x -k "ivf or search or robust" 2>&1 | tail -3
timeout 900 python tools/fuzz_ivf.py 777 60 2>&1 | tail -2
python tools/ivf_width_bench.py 96 2>&1 | grep -v amdgpu > gpurun_out/r6n/ivf_width_bench.txt; cat gpurun_out/r6n/ivf_width_bench.txt
