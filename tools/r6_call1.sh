#!/bin/bash
# round 6, GPU call 1: ubench table for the all-resident candidates, A/B of the q16 screen, GPU tests, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench_lds_gather.hip -o /tmp/ulds && /tmp/ulds 300 r6 > gpurun_out/r6a/ubench_r6.txt 2>&1
timeout 900 tools/adc_ab.sh run 32 48 > gpurun_out/r6a/adc_ab.txt 2>&1
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6a/pytest_gpu.txt
tail -3 gpurun_out/r6a/pytest_gpu.txt
