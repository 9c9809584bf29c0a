#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests -q -m gpu -x -k "ivf or image" 2>&1 | tail -3 > gpurun_out/r6k/pytest_ivf.txt
cat gpurun_out/r6k/pytest_ivf.txt
python tools/ivf_width_bench.py 96 > gpurun_out/r6k/ivf_width_bench.txt 2>&1; cat gpurun_out/r6k/ivf_width_bench.txt
python tools/ivf_width_bench.py 48 > gpurun_out/r6k/ivf_width_bench_m48.txt 2>&1; cat gpurun_out/r6k/ivf_width_bench_m48.txt
timeout 900 python tools/fuzz_ivf.py 4242 40 2>&1 | tail -2
