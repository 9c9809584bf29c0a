#!/bin/bash
# Round 6: A/B timing of the 16-query ADC screen under structural switches (development tool, GPU).
#   build here (no GPU):  tools/adc_ab.sh build         -> build/var/ab_<name>.so  (travels with gpurun)
#   on the GPU box:       tools/adc_ab.sh run [M ...]   -> one table per library (tools/adc_quick_bench.py)
# nosurv changes RESULTS (a timing probe of the kernel without its survivor path); the others are complete kernels.
# (Round 6 also timed per-gather waits and a kernel without its code loads: no gain / slower, profiles/r06a_adc_ab.txt; removed.)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
names=(nosurv late0 w12r12 w12r8 w8r16 pack12)
flags=("-DADC_EXP_NOSURV" "-DADC_Q16_LATE_TEST=0"
       "-DADC_Q16_WAVES=12 -DADC_Q16_R=12 -DADC_Q16_TILE=32256" "-DADC_Q16_WAVES=12 -DADC_Q16_R=8 -DADC_Q16_TILE=30720"
       "-DADC_Q16_WAVES=8 -DADC_Q16_R=16" "-DADC_Q16_PACK=1 -DADC_Q16_R=12 -DADC_Q16_TILE=30720")
if [ "$1" == "build" ]; then
  for i in "${!names[@]}"; do $root/tools/mkvar.sh ab_${names[$i]} adc_search.hip ${flags[$i]} & done
  wait
  exit 0
fi
shift || true
Ms=${@:-32 48}
echo "== lib: shipped"; python $root/tools/adc_quick_bench.py $Ms 2>&1 | grep "k=1000"
for n in "${names[@]}"; do
  echo "== lib: $n"
  REPCONC_HIP_LIB=$root/build/var/ab_$n.so python $root/tools/adc_quick_bench.py $Ms 2>&1 | grep "k=1000\|rror" | head -4
done
echo "== lib: shipped, RC_ADC_Q16_PRIO=0"; RC_ADC_Q16_PRIO=0 python $root/tools/adc_quick_bench.py $Ms 2>&1 | grep "k=1000"
echo "== lib: shipped"; python $root/tools/adc_quick_bench.py $Ms 2>&1 | grep "k=1000"
