#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests -q -m gpu -x -k "ivf" 2>&1 | tail -3 > gpurun_out/r6j/pytest_ivf.txt
cat gpurun_out/r6j/pytest_ivf.txt
python tools/ivf_width_bench.py 96 > gpurun_out/r6j/ivf_width_bench.txt 2>&1; cat gpurun_out/r6j/ivf_width_bench.txt
for a in "128 1200" "128 6980"; do REPCONC_HIP_LIB=$GRAFT_REPO_ROOT/build/var/trace16.so python tools/ivf16_timeline.py $a 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6j/ivf16_timeline_b.txt
cat gpurun_out/r6j/ivf16_timeline_b.txt
