cd $GRAFT_REPO_ROOT; ROOT=$GRAFT_REPO_ROOT
python tools/adc_quick_bench.py 48 2>&1 | grep "QPS"
for t in r6 r2; do echo $t; REPCONC_HIP_LIB=$ROOT/repconc_amd/lib/librepconc_hip_$t.so python tools/adc_quick_bench.py 48 2>&1 | grep "QPS"; done
