#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
tools/pmc_collect.sh gpurun_out/r6i/r06i_pmc_ivf16_nprobe128_raw.json -- python tools/ivf_one.py 128 > gpurun_out/r6i/pmc16.out 2>&1
RC_IVF_WIDTH=8 tools/pmc_collect.sh gpurun_out/r6i/r06i_pmc_ivf8_nprobe128_raw.json -- python tools/ivf_one.py 128 > gpurun_out/r6i/pmc8.out 2>&1
timeout 1500 python bench.py --steps 10 --warmup 3 > gpurun_out/r6i/bench.json 2> gpurun_out/r6i/bench.err
tail -c 1200 gpurun_out/r6i/bench.json
