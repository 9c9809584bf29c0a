#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
for a in "128 1200" "128 6980" "32 6980"; do REPCONC_HIP_LIB=$GRAFT_REPO_ROOT/build/var/trace16.so python tools/ivf16_timeline.py $a 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6j/ivf16_timeline.txt
cat gpurun_out/r6j/ivf16_timeline.txt
