cd $GRAFT_REPO_ROOT; O=gpurun_out/r02i; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -150 > $O/pytest_gpu.txt; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.txt | tail -20
python tools/warmup_bench.py 2>&1 | tail -5
