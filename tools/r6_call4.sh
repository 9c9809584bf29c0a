#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
R=$GRAFT_REPO_ROOT
( echo "== lib: shipped (flag store RELEASE, flag load ACQUIRE)"; python tools/xchg_bench.py 2>&1 | tail -9
  echo "== lib: relaxed flag store (round 5)"; REPCONC_HIP_LIB=$R/build/var/ab_relaxed.so python tools/xchg_bench.py 2>&1 | tail -9 ) > gpurun_out/r6d/xchg_bench.txt 2>&1
timeout 900 python tools/stage1_step_bench.py 512 11 3 > gpurun_out/r6d/stage1_step.txt 2> gpurun_out/r6d/stage1_step.err
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r6d/bench.json 2> gpurun_out/r6d/bench.err
timeout 600 env RC_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-adc --no-cpu > gpurun_out/r6d/bench_gpus2_shared.json 2> gpurun_out/r6d/bench_gpus2_shared.err
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6d/pytest_gpu.txt
tail -3 gpurun_out/r6d/pytest_gpu.txt; tail -3 gpurun_out/r6d/stage1_step.txt; tail -5 gpurun_out/r6d/stage1_step.err
