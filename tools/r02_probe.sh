#!/bin/bash
# round-2 GPU pass: new tests, sweep A/B, ADC A/B, PMC of the two hot kernels
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -k "sweep_variant or per_rank_shape or epsilon_outside or conflict_free or scan_image or constrained_codes_bit_exact or adc_integer or adc_search_matches or virtual_shards or native_rccl or sinkhorn_algorithm" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > $O/pytest_new.txt
tail -8 $O/pytest_new.txt
for B in 49152 6144; do
  RC_SK_V1=1 python tools/sweep_bench.py $B 48 40 2>&1 | tail -1 | sed 's/^/v1 default: /' >> $O/sweep.txt
  python tools/sweep_bench.py $B 48 40 2>&1 | tail -1 | sed 's/^/v2 fklds=1: /' >> $O/sweep.txt
  RC_SK_FKLDS=0 python tools/sweep_bench.py $B 48 40 2>&1 | tail -1 | sed 's/^/v2 fklds=0: /' >> $O/sweep.txt
  RC_SK_NB=2048 python tools/sweep_bench.py $B 48 40 2>&1 | tail -1 | sed 's/^/v2 nb=2048: /' >> $O/sweep.txt
  RC_SK_NB=512 python tools/sweep_bench.py $B 48 40 2>&1 | tail -1 | sed 's/^/v2 nb=512: /' >> $O/sweep.txt
done
cat $O/sweep.txt
python tools/adc_quick_bench.py 48 96 > $O/adc.txt 2>&1
grep "QPS" $O/adc.txt
python tools/config_bench.py 2>&1 | grep constrained > $O/config.txt; cat $O/config.txt
tools/pmc_collect.sh $O/pmc_adc.json -- python $ROOT/tools/adc_quick_bench.py 48 > $O/pmc_adc.out 2>&1
tools/pmc_collect.sh $O/pmc_sweep.json -- python $ROOT/tools/sweep_bench.py 49152 48 10 > $O/pmc_sweep.out 2>&1
python - <<PY
import json
for f,k in (("$O/pmc_adc.json","adc_screen_cf"),("$O/pmc_sweep.json","sk_sweep2")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, e); continue
    for name,v in d.items():
        if k in name:
            print(name[:60], {c: round(x["mean"]) for c,x in v.items()})
PY
