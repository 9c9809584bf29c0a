#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
tools/pmc_collect.sh gpurun_out/r6i/r06i_pmc_ivf16_nprobe128_raw.json -- python $R/tools/ivf_one.py 128 > gpurun_out/r6i/pmc16.out 2>&1
RC_IVF_WIDTH=8 tools/pmc_collect.sh gpurun_out/r6i/r06i_pmc_ivf8_nprobe128_raw.json -- python $R/tools/ivf_one.py 128 > gpurun_out/r6i/pmc8.out 2>&1
( echo "== lib: shipped (8-query screen requests the next stage's codes before its gathers)"; python tools/ivf_width_bench.py 96
  echo "== lib: codes requested after the last gather (rounds 3-5)"; REPCONC_HIP_LIB=$R/build/var/ab_ivf_nopf.so python tools/ivf_width_bench.py 96
  echo "== lib: shipped"; python tools/ivf_width_bench.py 96 ) > gpurun_out/r6i/ivf8_prefetch_ab.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -x -k "ivf" 2>&1 | tail -4 > gpurun_out/r6i/pytest_ivf.txt
cat gpurun_out/r6i/ivf8_prefetch_ab.txt; tail -3 gpurun_out/r6i/pytest_ivf.txt
