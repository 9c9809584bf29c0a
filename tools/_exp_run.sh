cd $GRAFT_REPO_ROOT; O=gpurun_out/r02f; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print(d.get("per_rank_6144"))
print(d["adc"]["value"], d["adc"]["roofline"]["frac"], d["adc"].get("other_shapes_queries_per_sec"))
print(d["index_build"]["value"], d["index_build"].get("cpu_baseline"))
print(d.get("cpu_baseline"), d.get("cpu_baseline_single_thread"), d.get("kmeans_stats"))
PY
