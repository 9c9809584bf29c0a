#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
( echo "# tools/fuzz_ivf.py (both screen widths forced in every trial), tools/fuzz_adc.py, tools/fuzz_nearest.py on HEAD"
  timeout 1500 python tools/fuzz_ivf.py 321 40 2>&1 | tail -3
  timeout 1500 python tools/fuzz_ivf.py 99 40 2>&1 | tail -3
  timeout 900 python tools/fuzz_adc.py 777 30 2>&1 | tail -2
  timeout 900 python tools/fuzz_nearest.py 2>&1 | tail -2 ) > gpurun_out/r6h/fuzz.txt 2>&1
timeout 900 env RC_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 2 --warmup 1 --no-adc --no-cpu > gpurun_out/r6h/bench_gpus8_shared.json 2> gpurun_out/r6h/bench_gpus8_shared.err
cat gpurun_out/r6h/fuzz.txt; tail -c 1500 gpurun_out/r6h/bench_gpus8_shared.json
