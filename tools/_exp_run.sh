cd $GRAFT_REPO_ROOT; ROOT=$GRAFT_REPO_ROOT
for i in 1 2; do
python tools/adc_quick_bench.py 96 2>&1 | grep "k=1000"
REPCONC_HIP_LIB=$ROOT/repconc_amd/lib/librepconc_hip_t64k.so python tools/adc_quick_bench.py 96 2>&1 | grep "k=1000"
done
