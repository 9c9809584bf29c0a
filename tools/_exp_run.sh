cd $GRAFT_REPO_ROOT; ROOT=$GRAFT_REPO_ROOT
python tools/adc_quick_bench.py 48 96 2>&1 | grep QPS
for T in 131072 262144; do echo tile $T; REPCONC_HIP_LIB=$ROOT/repconc_amd/lib/librepconc_hip_t$T.so python tools/adc_quick_bench.py 48 96 2>&1 | grep QPS; done
