cd $GRAFT_REPO_ROOT; O=gpurun_out/r02j; mkdir -p $O; ROOT=$GRAFT_REPO_ROOT
python tools/adc_quick_bench.py 48 96 > $O/adc.txt 2>&1; grep QPS $O/adc.txt
export REPCONC_HIP_LIB=$ROOT/repconc_amd/lib/librepconc_hip_dual.so
python tools/adc_quick_bench.py 48 96 > $O/adc_dual.txt 2>&1; grep QPS $O/adc_dual.txt
