#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for n in shipped late0 nosurv; do
  echo "== lib: $n"
  if [ $n == shipped ]; then python tools/adc_quick_bench.py 32 48 64 96 2>&1 | grep "k=1000";
  else REPCONC_HIP_LIB=$R/build/var/ab_$n.so python tools/adc_quick_bench.py 32 48 64 96 2>&1 | grep "k=1000"; fi
done
done > gpurun_out/r6b/adc_ab.txt 2>&1
timeout 1200 python -m pytest tests -q -m gpu -x -k "adc or search or ivf or index or sinkhorn_algorithm or abi" 2>&1 | tail -8 > gpurun_out/r6b/pytest_adc.txt
timeout 600 python tools/fuzz_adc.py 2>&1 | tail -5 > gpurun_out/r6b/fuzz_adc.txt
tail -3 gpurun_out/r6b/pytest_adc.txt; cat gpurun_out/r6b/adc_ab.txt
