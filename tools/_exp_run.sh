cd $GRAFT_REPO_ROOT; O=gpurun_out/r02e; mkdir -p $O; ROOT=$GRAFT_REPO_ROOT
python tools/adc_quick_bench.py 48 96 > $O/adc.txt 2>&1; grep QPS $O/adc.txt
export REPCONC_HIP_LIB=$ROOT/repconc_amd/lib/librepconc_hip_img16.so
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "conflict_free or scan_image or adc_integer or adc_large" 2>&1 | tail -5 > $O/pytest_adc16.txt; cat $O/pytest_adc16.txt
python tools/adc_quick_bench.py 48 96 64 32 > $O/adc16.txt 2>&1; grep QPS $O/adc16.txt
tools/pmc_collect.sh $O/pmc_adc16.json -- python $ROOT/tools/adc_quick_bench.py 48 > $O/pmc_adc.out 2>&1
python - <<PY
import json
d=json.load(open("$O/pmc_adc16.json"))
for name,v in d.items():
    if "adc_screen_cf" in name: print(name[:50], {c: round(x["mean"]) for c,x in v.items()})
PY
